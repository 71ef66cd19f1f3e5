#!/bin/bash
# Profiling recipe run on the GPU box (via gpurun).  Usage: bash tools/prof.sh <lattices> <tag>
# The stats pass profiles the default bench command (4 launches in flight in the timed region, then the serial pass: both grid
# sizes appear in the summary); the counter passes run one launch per step (--streams 1): counters serialise kernels anyway.
set -e
B=${1:-2048}; TAG=${2:-r02}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --lattices $B --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'forward_ck|backtrace_rc|forward_w16|backtrace_w16|gather_outputs|forward_tp|forward_ts|chunk_map|compose_maps|chain_entries' --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex 'forward_ck|backtrace_rc|forward_w16|backtrace_w16|gather_outputs|forward_tp|forward_ts|chunk_map|compose_maps|chain_entries' --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-include-regex 'forward_ck|backtrace_rc|forward_w16|backtrace_w16|gather_outputs|forward_tp|forward_ts|chunk_map|compose_maps|chain_entries' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/pmc_sq.err || echo "pmc_sq pass failed"
# the few-lattice forms (one cfg2 lattice: tile pipeline + chunk-parallel backtrace): traffic of the same kernels
RX='forward_tp|forward_ts|chunk_map|compose_maps|chain_entries|backtrace_rc|gather_outputs'
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --kernel-trace --output-format csv -d $OUT/pmc_fetch_single -- python3 $R/bench.py --lattices 1 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/pmc_fetch_single.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --kernel-trace --output-format csv -d $OUT/pmc_write_single -- python3 $R/bench.py --lattices 1 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/pmc_write_single.err
find $OUT -name "*.csv" | head -40
