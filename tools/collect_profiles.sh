#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats and three separate PMC passes of the
# default bench (two-fluid Orszag-Tang 4096^2) and the 3-D workload.  Output: gpurun_out/prof_*/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
B3="python3 $R/bench.py --workload rmi_3d --cells 256 --steps 4 --warmup 1 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ot_stats -o ot -- $B > $R/gpurun_out/prof_ot_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_ot_fetch -o ot -- $B > $R/gpurun_out/prof_ot_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_ot_write -o ot -- $B > $R/gpurun_out/prof_ot_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_ot_sq -o ot -- $B > $R/gpurun_out/prof_ot_sq.log 2>&1
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rmi_stats -o rmi -- $B3 > $R/gpurun_out/prof_rmi_stats.log 2>&1
echo done
