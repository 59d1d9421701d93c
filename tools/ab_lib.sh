#!/bin/bash
# diagnostic: A/B of two builds of the library on one box (box-to-box clock differences are a few %):
#   tools/ab_lib.sh tools/_lib_a.so tools/_lib_b.so ["bench args"]   (alternates them; restores the first at the end)
A=$1; B=$2; ARGS=${3:-"--steps 30 --reps 3"}
for rep in 1 2; do
  for L in $A $B; do
    cp $L cerberus_amd/libcerberus_mfp.so
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline $ARGS > gpurun_out/ab.json 2>/dev/null || exit 1
    python - "$L" <<PY
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], "%.3f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], {a: round(b, 4) for a, b in d["kernel_ms_per_step"].items()})
PY
  done
done
cp $A cerberus_amd/libcerberus_mfp.so
