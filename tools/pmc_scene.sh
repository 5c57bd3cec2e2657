#!/bin/bash
# instruction and wait counters of the tracer kernels of one scene: bash tools/pmc_scene.sh config/cbox.conf 1024 64 f64
R=$(pwd)
O=$R/gpurun_out/pmc_scene
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $O -o pmc -- python3 $R/tools/time_scene.py "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
grep -E "Mpaths" $O/run.log
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
    if "ptx::" in k and c.get("SQ_INSTS_VALU", 0) > 1e6:
        print("%-40s launches %3d VALU %.4g SALU %.4g LDS %.4g lanes/inst %.3f wait %.3f" % (k[-40:], len(n[k]), c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"],
              c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1) / 64, c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1)))
PY
