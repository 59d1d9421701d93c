#!/bin/bash
# diagnostic: z-segment sweep of the 3-D hydro marching kernel; usage: tools/seg_sweep3d.sh N "ENV=VAL ..." ...
n=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --workload rmi_3d --n $n --steps 10 --reps 3 > gpurun_out/sw3.json 2>/dev/null || exit 1
  python - "$e" <<PY
import json, sys
d = json.load(open("gpurun_out/sw3.json"))
print(sys.argv[1], "%.3f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], {a: round(b, 4) for a, b in d["kernel_ms_per_step"].items()})
PY
done
