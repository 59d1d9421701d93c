#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_*) into the
summaries committed under profiles/<round>/: kernel-trace stats, per-kernel PMC averages and the
HBM traffic per launch of the fused kernels (gfx950 corrections of MI355X_MICROARCH.md: the
counters are KiB; FETCH_SIZE tallies 128-B read requests at 64 B, so read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact).  usage: summarize_profiles.py <round dir, e.g. profiles/r01> [n]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
os.makedirs(out, exist_ok=True)


def first(pattern):
    f = glob.glob(os.path.join(go, pattern), recursive=True)
    return f[0] if f else None


for src, dst in (("prof_ot_stats/**/*kernel_stats.csv", f"kernel_stats_ot2f_{n}.csv"),
                 ("prof_rmi_stats/**/*kernel_stats.csv", "kernel_stats_rmi3d_256.csv")):
    f = first(src)
    if f:
        shutil.copy(f, os.path.join(out, dst))

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("prof_ot_fetch", "prof_ot_write", "prof_ot_sq"):
    f = first(d + "/**/*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
avg = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
json.dump(avg, open(os.path.join(out, f"pmc_per_kernel_avg_ot2f_{n}.json"), "w"), indent=1, sort_keys=True)

cells = n * n


def pick(sub):
    ks = [k for k in avg if sub in k and "FETCH_SIZE" in avg[k]]
    return max(ks, key=lambda k: avg[k]["FETCH_SIZE"]) if ks else None


traffic = {"_comment": "HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE), gfx950 "
                       "corrections of MI355X_MICROARCH.md: KiB units, read bytes = FETCH_SIZE*1024*2, write bytes = "
                       "WRITE_SIZE*1024.  Written by tools/summarize_profiles.py."}
entry = {}
for tag, sub, ncomp in (("ctu_hydro", "k_ctu_march<", 5), ("ctu_hydro_tiled", "k_ctu2d<fast::HydroF", 5),
                        ("ctu_field", "k_ctu2d<fast::FieldF", 10), ("source_rk2", "k_source_rk2<2, false, ", 20)):
    k = pick(sub)
    if not k:
        continue
    rd, wr = avg[k]["FETCH_SIZE"] * 1024 * 2, avg[k].get("WRITE_SIZE", 0.0) * 1024
    entry[tag + "_kernel"] = k[:90]
    entry[tag + "_fetch_bytes"] = int(rd)
    entry[tag + "_write_bytes"] = int(wr)
    entry[tag + "_bytes_per_launch"] = int(rd + wr)
    if tag == "ctu_field" and ", 2>(" in k:   # field update with the source epilogue of two species: all 20 components
        ncomp = 20
    entry[tag + "_algorithmic_bytes"] = 2 * 8 * ncomp * cells
traffic[f"ot2f_2d:{n}"] = entry
json.dump(traffic, open(os.path.join(os.path.dirname(out.rstrip("/")), os.path.basename(out.rstrip("/")) + "_traffic.json"), "w"), indent=2)
print(json.dumps(entry, indent=1))
