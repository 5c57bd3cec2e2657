#!/bin/bash
# wavefront tracer on room and mis under different options: bash tools/sweep_wavefront.sh  (GPU box, repository root)
set -e
for scene in room mis; do
for mode in f64 f32; do
for opts in "bvh_wavefront=1" "wf_refill=8" "wf_refill=40" "lds_nodes=0" "lds_stack_levels=24" "lds_stack_levels=12" "trace_blocks_per_cu=4" "trace_blocks_per_cu=6" "default_batch_paths=16777216" "default_batch_paths=8388608" "default_batch_paths=4194304"; do
  echo "== $scene $mode $opts"
  PTX_DEBUG=1 PTX_OPTS=$opts timeout -k 10 120 python3 tools/time_scene.py config/$scene.conf 1024 32 $mode 2>&1 | grep -E "Mpaths|trace kernel" | sort -u
done; done; done
