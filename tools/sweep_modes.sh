for n in 128 256 512 1024 2048; do for m in workgroup wave; do timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --lattices $n --mode $m > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$n $m failed"; continue; }; python - "$n" "$m" <<PY
import json,sys
d=json.load(open("gpurun_out/sw.json")); k=d["kernels_ms"]
print(sys.argv[1], sys.argv[2], "ms/step %.2f fwd %.2f bt %.2f ga %.2f" % (d["ms_per_step"], k["forward"], k["backtrace"], k["gather"]))
PY
done; done
