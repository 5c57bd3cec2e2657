#!/bin/bash
# runs the fetch variants on the GPU box: bash tools/microbench/run.sh
cd $(dirname $0)
for act in 64 36; do for wps in 3 5; do for v in 0 1 2; do
  timeout -k 5 60 ./node_fetch $v 136000 2000 $act $wps || exit 1
done; done; done
for v in 0 1 2; do timeout -k 5 60 ./node_fetch $v 8000 2000 36 3 || exit 1; done
