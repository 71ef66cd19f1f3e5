#!/bin/bash
# Instruction mix and wait counters of ONE tile wavefront (a one-tile lattice: T=50000, S=100) in the tiled form and of the
# one-wavefront form on the same lattice (run on the GPU box):  bash tools/pmc_tile.sh <tag>
TAG=${1:-tile}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/one.py <<PY
import sys
sys.path.insert(0, "$R")
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
for mode in ("tiled", "wave"):
    lps, labs = W.device_book([(50000, 100)], V=64, seed0=0)
    b = DeviceBatch(lps, labs, 1000)
    b.engine.set_mode(mode)
    b.run()
    b.engine.set_mode("auto")
PY
cd /tmp && export TMPDIR=/tmp
RE='forward_tp|forward_ck'
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $OUT/one.py > /dev/null 2> $OUT/p1.err || echo "p1 failed"
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $OUT/one.py > /dev/null 2> $OUT/p2.err || echo "p2 failed"
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p3 -- python3 $OUT/one.py > /dev/null 2> $OUT/p3.err || echo "p3 failed"
rocprofv3 --kernel-include-regex "$RE" --pmc SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p4 -- python3 $OUT/one.py > /dev/null 2> $OUT/p4.err || echo "p4 failed"
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(dict)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:34]
        v=float(r['Counter_Value'])
        agg[k][r['Counter_Name']]=max(agg[k].get(r['Counter_Name'],0),v)   # the instance that did the work
for k,v in agg.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-24s %12.4g  per frame %.2f'%(c,x,x/50000))
PY
