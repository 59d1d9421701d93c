#!/bin/bash
# SQ counter pass of a bench command: tools/pmc_sq.sh <tag> <bench args...>; output gpurun_out/pmc_<tag>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc ${PMC:-SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY} --output-format csv -d $R/gpurun_out/pmc_$TAG -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:70]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,v in acc.items():
    if "ctu" in k or "source" in k or "g_flux" in k or "march" in k or "f3d" in k:
        w=v["SQ_WAVES"]; print(k, "launches",cnt[k], {n:round(x/w,1) for n,x in v.items() if n!="SQ_WAVES"}, "waves/launch", w/cnt[k])
PY
