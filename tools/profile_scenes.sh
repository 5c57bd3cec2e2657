#!/bin/bash
# Bench lines of the three scenes plus a kernel trace of the BVH scenes on the GPU box.
# usage: bash tools/profile_scenes.sh <tag>        (run from the repository root)
# Writes gpurun_out/scenes_<tag>/: bench_cbox.json (default bench.py, with the CPU baseline),
# bench_room.json, bench_mis.json, room_kernel_stats.csv, mis_kernel_stats.csv.
set -e
TAG=${1:-r01_v9}
R=$(pwd)
O=$R/gpurun_out/scenes_$TAG
mkdir -p $O
line() { grep '^{"metric"' "$1" | tail -1; }
timeout -k 10 300 python3 bench.py > $O/cbox.log 2>&1
line $O/cbox.log > $O/bench_cbox.json
timeout -k 10 300 python3 bench.py --conf config/room.conf --spp 64 --steps 2 --warmup 1 > $O/room.log 2>&1
line $O/room.log > $O/bench_room.json
timeout -k 10 300 python3 bench.py --conf config/mis.conf --spp 256 --steps 2 --warmup 1 --cpu-seconds 8 > $O/mis.log 2>&1
line $O/mis.log > $O/bench_mis.json
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_room -o kt -- python3 $R/bench.py --conf config/room.conf --spp 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/room_kt.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_mis -o kt -- python3 $R/bench.py --conf config/mis.conf --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/mis_kt.log 2>&1
cd $R
python3 tools/rocpd_summary.py $(find $O/kt_room -name "*.db" | head -1) > $O/room_kernel_stats.csv
python3 tools/rocpd_summary.py $(find $O/kt_mis -name "*.db" | head -1) > $O/mis_kernel_stats.csv
line $O/room_kt.log > $O/bench_room_profiled.json
line $O/mis_kt.log > $O/bench_mis_profiled.json
find $O -name "*.db" -delete
head -4 $O/room_kernel_stats.csv $O/mis_kernel_stats.csv
cut -c1-200 $O/bench_cbox.json $O/bench_room.json $O/bench_mis.json
