#!/bin/bash
# instruction counters of the physically based estimator's kernels (GPU box): bash tools/pmc_fast.sh <bench args>
R=$(pwd); O=$R/gpurun_out/pmc_fast; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scenes --estimator fast "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, c in acc.items():
    if "ptx::" in k and c.get("SQ_INSTS_VALU", 0) > 1e8:
        L = len(n[k])
        print("%-40s launches %3d per launch: VALU %.4g SALU %.4g wavecyc %.4g  lanes/inst %.3f wait %.3f waitinst %.3f" % (k[-40:], L, c["SQ_INSTS_VALU"] / L, c["SQ_INSTS_SALU"] / L, c["SQ_WAVE_CYCLES"] / L,
              c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1) / 64, c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1)))
PY
find $O -name "*.csv" -size +20M -delete
