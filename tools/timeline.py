#!/usr/bin/env python3
"""Kernel timeline from a rocprofv3 --kernel-trace rocpd database.  usage: timeline.py results.db [first] [last]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
rows = list(cur.execute(f"select kernel_id,start,end,queue_id,grid_size_x from {kd} order by start"))
t0 = rows[0][1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
last = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
for r in rows[first:last]:
    nm = names[r[0]].split('(')[0].replace('void ptx::', '')
    if 'rocclr' in nm:
        continue
    print("%-34s start %9.3f ms  end %9.3f  dur %8.3f ms  queue %s" % (nm[:34], (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3]))
