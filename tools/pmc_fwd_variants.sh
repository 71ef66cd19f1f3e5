#!/bin/bash
# SQ instruction counters of the forward kernel for library variants (build_variants/lib<NAME>.so), one launch of 8192 lattices:
#   gpurun -- 'bash tools/pmc_fwd_variants.sh m0 kill'
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export KA_LIBRARY=$R/build_variants/lib$v.so
  OUT=$R/gpurun_out/pmc_fwd_$v; rm -rf $OUT
  rocprofv3 --kernel-include-regex 'forward_ck' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --lattices 8192 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT.err || echo "$v failed"
  python3 - "$v" "$OUT" <<'PY'
import csv, glob, sys, collections
v, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = 0
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "forward_ck" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 8192 * 64:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n += 1
frames = 8192 * 50000
d = len(set(1 for _ in [0]))  # placeholder
print(v, {k: round(x, 1) for k, x in acc.items()})
disp = max(1, n // max(1, len(acc)))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
    if k in acc: print("  ", k, "per frame and wavefront:", round(acc[k] / disp / frames, 2), "(dispatches", disp, ")")
PY
done
