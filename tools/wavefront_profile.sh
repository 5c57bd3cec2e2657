#!/bin/bash
# per-kernel time of tools/wavefront_check.py (rocprofv3 kernel trace): bash tools/wavefront_profile.sh config/room.conf 1024 32 f64
R=$(pwd)
O=$R/gpurun_out/wf_prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o wf -- python3 $R/tools/wavefront_check.py "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
grep -E "Mpaths|largest" $O/run.log
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-60s calls %6s total %10.3f ms avg %9.3f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
