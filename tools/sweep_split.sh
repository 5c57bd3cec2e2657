#!/bin/bash
# reference-splitting budget against throughput and records visited per ray (GPU box)
for conf in config/room.conf config/mis.conf; do
for v in ${SPLITS:-0 1 2 5 10}; do
  echo "== $conf bvh_split=$v"
  PTX_OPTS=bvh_split=$v timeout -k 5 120 python3 tools/time_scene.py $conf 1024 32 f64 || exit 1
done; done
