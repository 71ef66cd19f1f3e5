#!/bin/bash
# Build the library and its assembly, run the in-flight-register lint; non-zero exit when anything fails.
# Run this before every gpurun:  tools/preflight.sh && gpurun ...
set -e
cd "$(dirname "$0")/.."
out=$(make -C kokoro-align_amd/csrc 2>&1) || { echo "$out" | grep -E "error" | head -20; echo "BUILD FAILED"; exit 1; }
make -C kokoro-align_amd/csrc asm >/dev/null 2>&1 || { echo "ASM BUILD FAILED"; exit 1; }
python tools/lint_inflight.py | tail -2
python tools/lint_inflight.py >/dev/null
