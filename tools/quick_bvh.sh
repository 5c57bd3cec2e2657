#!/bin/bash
# room and mis, both precisions, reference estimator and the physically based one at 1024 x 1024 x 32 spp (GPU box)
for conf in config/room.conf config/mis.conf; do for mode in f64 f32; do
  timeout -k 5 120 python3 tools/time_scene.py $conf 1024 32 $mode 2>&1 | grep -E "Mpaths|BVH records" || exit 1
done; done
