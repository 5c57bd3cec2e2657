#!/bin/bash
# Issue-cycle accounting of a tracer kernel on the GPU box: the opcode-cost micro-benchmark, then three
# counter passes over the bench command (per-class VALU instruction counts, f32 / integer side; f64
# side and the other instruction kinds; busy and wait cycles), each in a run of its own; then
# tools/issue_cycles.py dumps the raw counters (and writes a first profiles/<tag>_issue_cycles.json without the static mix).
# usage: bash tools/issue_cycles.sh <tag> [bench.py arguments]      (run from the repository root)
set -e
TAG=${1:-r03_cbox}
shift || true
R=$(pwd)
O=$R/gpurun_out/issue_$TAG
mkdir -p $O
timeout -k 5 120 $R/tools/microbench/valu_issue 2000 > $O/valu_issue.json
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fast-estimator --no-scenes $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/bench_kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/a -o a -- $CMD > $O/bench_a.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $O/b -o b -- $CMD > $O/bench_b.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $O/c -o c -- $CMD > $O/bench_c.log 2>&1
cd $R
grep '^{"metric"' $O/bench_kt.log | tail -1 > $O/bench.json
python3 tools/issue_cycles.py dump $(find $O/kt -name "*.db" | head -1) $(find $O/a -name "*.db" | head -1) $(find $O/b -name "*.db" | head -1) $(find $O/c -name "*.db" | head -1) > $O/raw_counters.json
find $O -name "*.db" -delete
# back in the build container (the assembly is not on the GPU box):
#   python3 tools/issue_cycles.py combine $TAG $O/valu_issue.json $O/bench.json $O/raw_counters.json --isa context.s
python3 tools/issue_cycles.py combine $TAG $O/valu_issue.json $O/bench.json $O/raw_counters.json
