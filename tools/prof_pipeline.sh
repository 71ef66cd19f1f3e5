set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe2; rm -rf $O; mkdir -p $O
cd $R
python tools/bench_pipeline.py 64 43000 128 device 2>/dev/null | tail -1 > $O/p64.json
python tools/bench_pipeline.py 64 43000 128 device_steps 2>/dev/null | tail -1 > $O/s64.json
python tools/bench_pipeline.py 120 43000 128 device 2>/dev/null | tail -1 > $O/p120.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_pipeline.py 64 43000 128 device > $O/under_rocprof.json 2> $O/stats.err
find $O -name "*kernel_stats.csv" | head
