#!/bin/bash
# A/B timing of kernel variants on the GPU box.  Build each variant to build_variants/lib<NAME>.so (the directory is
# git-ignored but travels with gpurun), then:   gpurun -- './tools/ab_run.sh A B A B'
# Each run is the default bench without the CPU baseline; KA_LIBRARY selects the shared library.
mkdir -p gpurun_out
for v in "$@"; do
  KA_LIBRARY=$PWD/build_variants/lib$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 > gpurun_out/bv_$v.json 2> gpurun_out/bv_$v.err || { echo "$v failed"; tail -3 gpurun_out/bv_$v.err; continue; }
  python - "$v" <<PY
import json,sys
d=json.load(open(f"gpurun_out/bv_{sys.argv[1]}.json")); k=d["kernels_ms"]
print(sys.argv[1], "ms/step %.2f fwd %.2f bt %.2f ga %.2f ok=%s single=%.2f" % (d["ms_per_step"], k["forward"], k["backtrace"], k["gather"], d["parity_spot_check"], d["single_lattice"]["ms"]))
PY
done
