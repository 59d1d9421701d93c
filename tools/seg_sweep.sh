#!/bin/bash
# diagnostic: segment-length sweep of the marching kernels at one block size; usage: tools/seg_sweep.sh N "ENV=VAL ..." ...
n=$1; shift
for e in "$@"; do
  env $e timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --n $n --steps 50 --reps 3 > gpurun_out/sw.json 2>/dev/null || exit 1
  python - "$e" <<PY
import json, sys
d = json.load(open("gpurun_out/sw.json"))
print(sys.argv[1], "%.3f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], {a: round(b, 4) for a, b in d["kernel_ms_per_step"].items()})
PY
done
