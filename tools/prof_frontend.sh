#!/bin/bash
# Kernel statistics of the device front end on the GPU box:  bash tools/prof_frontend.sh [hours]
set -e
H=${1:-8.2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/frontend_prof; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_frontend.py $H 30 > $O/under_rocprof.json 2> $O/stats.err
find $O -name "*kernel_stats.csv" | head -2
