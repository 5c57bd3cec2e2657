#!/bin/bash
# room and mis through the ring tracer with different splits of the LDS between stack levels and node records
# usage (GPU box, repository root): bash tools/sweep_lds_nodes.sh
set -e
for scene in room mis; do
for mode in f64 f32; do
for opts in "lds_nodes=0" "lds_nodes=-1" "lds_nodes=-1,lds_stack_levels=8" "lds_nodes=-1,lds_stack_levels=12" "lds_nodes=-1,lds_stack_levels=20" "lds_nodes=64" "lds_nodes=128"; do
  echo "== $scene $mode $opts"
  PTX_DEBUG=1 PTX_OPTS=$opts timeout -k 10 120 python3 tools/time_scene.py config/$scene.conf 1024 32 $mode 2>&1 | grep -E "Mpaths|trace kernel" | sort -u
done; done; done
