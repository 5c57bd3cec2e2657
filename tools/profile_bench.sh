#!/bin/bash
# One profiling session of the headline bench command on the GPU box: kernel trace + three counter
# passes (each in a run of its own), then tools/make_profiles.py writes the summaries under profiles/.
# usage: bash tools/profile_bench.sh <tag> [scene [bench.py arguments]]      (run from the repository root)
# e.g.   bash tools/profile_bench.sh r01_room_v9 room --conf config/room.conf --spp 64
set -e
TAG=${1:-r01_v4}
R=$(pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SCENE=${2:-cbox}
shift || true
shift || true
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast-estimator --no-scenes $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/bench_kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/sq -o sq -- $CMD > $O/bench_sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- $CMD > $O/bench_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- $CMD > $O/bench_write.log 2>&1
cd $R
grep '^{"metric"' $O/bench_kt.log | tail -1 > $O/bench.json
python3 tools/make_profiles.py $TAG $O/bench.json $(find $O/kt -name "*.db" | head -1) $(find $O/sq -name "*.db" | head -1) $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) $SCENE
CAL=calibration.json
[ "$SCENE" = cbox ] || CAL=calibration_$SCENE.json
mkdir -p $R/gpurun_out/profiles_$TAG
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc.json profiles/$CAL $R/gpurun_out/profiles_$TAG/
find $O -name "*.db" -delete
cat profiles/${TAG}_kernel_stats.csv
