#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats and three separate PMC passes (FETCH_SIZE;
# WRITE_SIZE; an SQ set) of one bench command.   tools/collect_one.sh <tag> <bench.py args...>
# Output: gpurun_out/prof_<tag>_{stats,fetch,write,sq}/ ; tools/summarize_one.py turns them into profiles/<round>/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extras $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -o p -- $B > $R/gpurun_out/prof_${TAG}_stats.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -o p -- $B > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -o p -- $B > $R/gpurun_out/prof_${TAG}_write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -o p -- $B > $R/gpurun_out/prof_${TAG}_sq.log 2>&1 || exit 1
echo "collected $TAG"
