#!/bin/bash
# Instruction-mix counters of the batch kernels (run on the GPU box):  bash tools/pmc_sq.sh <lattices> <tag>
B=${1:-8192}; TAG=${2:-sq}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RE='forward_ck|backtrace_rc'
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/p1.err || echo "p1 failed"
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/p2.err || echo "p2 failed"
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p3 -- python3 $R/bench.py --lattices $B --steps 1 --warmup 0 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/p3.err || echo "p3 failed"
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(dict)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        if float(r['Counter_Value'])<1e6: continue
        agg[k][r['Counter_Name']]=agg[k].get(r['Counter_Name'],0)+float(r['Counter_Value'])
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-24s %.4g  per frame-wave %.2f'%(c,x,x/(8192*50000)))
PY
