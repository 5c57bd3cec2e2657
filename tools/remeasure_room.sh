#!/bin/bash
# Everything that is measured on the room scene, in one GPU session (after the scene changed):
# bench line with the CPU baseline, kernel trace + counter passes, the 3840x2160 frame.
# usage: bash tools/remeasure_room.sh <tag>       (run from the repository root)
set -e
TAG=${1:-r01_room_v10}
O=gpurun_out/room_$TAG
mkdir -p $O
timeout -k 10 300 python3 bench.py --conf config/room.conf --spp 64 --steps 2 --warmup 1 > $O/room.log 2>&1
grep '^{"metric"' $O/room.log | tail -1 > $O/bench_room.json
bash tools/profile_bench.sh $TAG room --conf config/room.conf --spp 64 > $O/prof.log 2>&1
timeout -k 10 400 python3 bench.py --conf config/room.conf --width 3840 --height 2160 --spp 32 --steps 2 --warmup 1 --cpu-seconds 10 > $O/room4k.log 2>&1
grep '^{"metric"' $O/room4k.log | tail -1 > $O/bench_room4k.json
cut -c1-160 $O/bench_room.json $O/bench_room4k.json
