set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_room
mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU -d $O/sq -o sq -- python3 $R/tools/time_scene.py config/room.conf 1024 16 f64 > $O/sq.log 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $O/tcp -o tcp -- python3 $R/tools/time_scene.py config/room.conf 1024 16 f64 > $O/tcp.log 2>&1 &&
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum -d $O/tcc -o tcc -- python3 $R/tools/time_scene.py config/room.conf 1024 16 f64 > $O/tcc.log 2>&1 &&
for p in sq tcp tcc; do echo "== $p"; python3 $R/tools/pmc_summary.py $(find $O/$p -name "*.db" | head -1) | grep -A12 k_trace; done > $O/summary.txt 2>&1
tail -3 $O/sq.log
cat $O/summary.txt
find $O -name "*.db" -delete
