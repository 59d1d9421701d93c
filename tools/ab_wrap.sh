for args in "--workload gem_2d --n 4096" "--n 2048" "--n 3072"; do
for cfg in "CMFP_MARCH_WRAP=0" "CMFP_MARCH_WRAP=4" "CMFP_MARCH_WRAP=1"; do
env $cfg timeout -k 10 100 python bench.py --no-extras --no-cpu-baseline --steps 30 --reps 3 $args > gpurun_out/ab.json 2>/dev/null || exit 1
python - <<PY
import json
d = json.load(open("gpurun_out/ab.json"))
print("$args", "$cfg", "%.3f G" % (d["value"] / 1e9), "%.4f ms" % d["ms_per_step"], {a: round(b, 4) for a, b in d["kernel_ms_per_step"].items()})
PY
done; done
