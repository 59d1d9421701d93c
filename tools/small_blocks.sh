#!/bin/bash
# throughput of the headline workload on small per-GPU blocks; usage: tools/small_blocks.sh [env assignments...]
for n in 512 1024 2048 4096; do
  env "$@" timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --n $n --steps 50 --reps 3 > gpurun_out/sb_$n.json 2>/dev/null || exit 1
  python - <<PY
import json
d = json.load(open("gpurun_out/sb_$n.json"))
k = d["kernel_ms_per_step"]
print($n, "%.3f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], {a: round(b, 4) for a, b in k.items()})
PY
done
