#!/bin/bash
# HBM read-traffic pass (FETCH_SIZE) of a bench command: tools/pmc_fetch.sh <tag> <bench args...>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc ${PMC:-FETCH_SIZE} --output-format csv -d $R/gpurun_out/pmcf_$TAG -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmcf_$TAG.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmcf_$TAG/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k,v in acc.items():
    if "ctu" in k or "source" in k or "g_flux" in k or "march" in k or "f3d" in k:
        print(k, {n:(round(x/cnt[k][n],1)) for n,x in v.items()}, "per launch (raw counter units)")
PY
