#!/bin/bash
# kernel trace of the physically based estimator on a scene (GPU box): bash tools/kt_fast.sh <tag> <bench args>
TAG=$1; shift
R=$(pwd); O=$R/gpurun_out/kt_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scenes --estimator fast "$@" > $O/bench.log 2>&1
cd $R
python3 - $O <<'PY'
import sys, glob
sys.path.insert(0, "tools")
from make_profiles import kernel_stats, short
db = glob.glob(sys.argv[1] + "/kt/**/*.db", recursive=True)[0]
for n, calls, tot, avg, mn, mx, pct in kernel_stats(db)[:9]:
    print("%-44s calls %3d  total %8.2f ms  avg %7.3f ms  %5.1f %%" % (short(n).replace("ptx::", "")[:44], calls, tot / 1e6, avg / 1e6, pct))
PY
find $O -name "*.db" -delete
