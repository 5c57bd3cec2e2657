#!/bin/bash
# Vector-memory counters of one bench step (texture addresser, vector L1, its address translation, L2):
# four counter passes, each a run of its own, summed per kernel into gpurun_out/cc_<tag>/summary.txt.
# usage: bash tools/cache_counters.sh <tag> [bench.py arguments]        (run from the repository root)
# e.g.   bash tools/cache_counters.sh room --conf config/room.conf --spp 16
set -e
TAG=${1:-room}
shift || true
R=$(pwd)
O=$R/gpurun_out/cc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
timeout -s KILL 200 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE -d $O/p1 -o p1 -- $CMD > $O/p1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $O/p2 -o p2 -- $CMD > $O/p2.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum -d $O/p3 -o p3 -- $CMD > $O/p3.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum -d $O/p4 -o p4 -- $CMD > $O/p4.log 2>&1
cd $R
: > $O/summary.txt
for p in p1 p2 p3 p4; do
    python3 tools/pmc_summary.py $(find $O/$p -name "*.db" | head -1) >> $O/summary.txt
done
find $O -name "*.db" -delete
cat $O/summary.txt
