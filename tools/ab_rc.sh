#!/bin/bash
# A/B timing of library variants (build_variants/lib<NAME>.so) on the batch bench only:
#   gpurun -- './tools/ab_rc.sh A B:7168 ..'      (NAME[:lattices])
mkdir -p gpurun_out
for a in "$@"; do
  v=${a%%:*}; n=${a#*:}; [ "$n" = "$a" ] && n=8192
  KA_LIBRARY=$PWD/build_variants/lib$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --steps 3 --lattices $n > gpurun_out/bv_${v}_$n.json 2> gpurun_out/bv_${v}_$n.err || { echo "$v failed"; tail -3 gpurun_out/bv_${v}_$n.err; continue; }
  python - "$v" "$n" <<PY
import json,sys
d=json.loads(open(f"gpurun_out/bv_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1]); k=d["kernels_ms"]
print(sys.argv[1], sys.argv[2], "Gframes/s %.3f ms/step %.2f fwd %.2f bt %.2f ga %.2f ok=%s" % (d["value"]/1e9, d["ms_per_step"], k["forward"], k["backtrace"], k["gather"], d["parity_spot_check"]))
PY
done
