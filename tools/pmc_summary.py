#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd database.  usage: pmc_summary.py results.db"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
ki, ci, vi = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
di = cols.index("dispatch_id") if "dispatch_id" in cols else None
seen = set()
for r in cur.execute("select * from counters_collection"):
    name = r[ki].split("(")[0][-60:]
    acc[name][r[ci]] += r[vi]
    if di is not None and (r[di], r[ci]) not in seen:
        seen.add((r[di], r[ci]))
for name, c in acc.items():
    print(name)
    for k, v in sorted(c.items()):
        print("   %-28s %.4g" % (k, v))
