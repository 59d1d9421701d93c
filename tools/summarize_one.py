#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_one.sh <tag> (gpurun_out/prof_<tag>_*) into the summaries
committed under profiles/<round>/: kernel_stats_<tag>.csv and pmc_<tag>.json = per-kernel, per-launch averages of
the counters plus HBM traffic with the gfx950 corrections of MI355X_MICROARCH.md (counters in KiB; FETCH_SIZE
tallies 128-B read requests at 64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact) and the
VALU-busy fraction = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES per SE summed ...) is left to the reader: the raw
SQ averages are kept.        usage: summarize_one.py <round dir> <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
os.makedirs(out, exist_ok=True)


def first(pattern):
    f = glob.glob(os.path.join(go, pattern), recursive=True)
    return f[0] if f else None


f = first(f"prof_{tag}_stats/**/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(out, f"kernel_stats_{tag}.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("fetch", "write", "sq"):
    f = first(f"prof_{tag}_{d}/**/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
dur = {}   # average launch duration (ns) per kernel from the kernel-trace pass
f = first(f"prof_{tag}_stats/**/*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"])
res = {}
for k in acc:
    if not any(s in k for s in ("k_ctu", "k_f3d", "k_f3m", "k_source", "k_g_", "k_fill", "k_pack", "k_proj", "k_fld", "k_reduce", "k_new_dt", "k_ell")):
        continue
    e = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    e["launches_seen"] = max(cnt[k].values())
    if "FETCH_SIZE" in e:
        e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in e:
        e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    if "SQ_WAVES" in e and e["SQ_WAVES"]:
        e["valu_insts_per_wave"] = e.get("SQ_INSTS_VALU", 0) / e["SQ_WAVES"]
    if k in dur:
        e["avg_launch_ns"] = dur[k]
        if "SQ_ACTIVE_INST_VALU" in e:
            # VALU-busy fraction of the launch: the counter advances once per 4 cycles a SIMD issues VALU work;
            # 256 CUs x 4 SIMDs at 2.4 GHz (the recipe of profiles/r01/README.md)
            e["valu_busy"] = e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * dur[k] * 2.4)
    res[k[:140]] = e
json.dump(res, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)):
    print(k[:100], "GB/launch", round(e.get("hbm_bytes_per_launch", 0) / 1e9, 3), "valu/wave", round(e.get("valu_insts_per_wave", 0)),
          "ms", round(e.get("avg_launch_ns", 0) / 1e6, 3), "valu_busy", round(e.get("valu_busy", 0), 2))
