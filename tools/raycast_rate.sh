#!/bin/bash
# kernel durations of tools/raycast_rate.py, one line per traversal kernel call, in call order
# usage (GPU box, repository root): bash tools/raycast_rate.sh config/room.conf "raycast_wavefront=1,wf_refill=24;..."
R=$(pwd)
O=$R/gpurun_out/raycast_rate
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PTX_DEBUG=1 timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $O -o rr -- python3 $R/tools/raycast_rate.py $1 --variants "$2" $3 $4 $5 $6 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
grep -E "operator|wavefront|differ|sorted" $O/run.log | sort -u
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_raycast" in n or "k_wf_traverse" in n:
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        print("%-40s grid %8s  %.3f ms" % (n[:40], r.get("Grid_Size", "?"), ms))
PY
