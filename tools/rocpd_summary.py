#!/usr/bin/env python3
"""Dumps the per-kernel statistics of a rocprofv3 run (rocpd SQLite output, `--kernel-trace
--stats`) as CSV:  name, calls, total_us, avg_us, percent.   usage: rocpd_summary.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("name,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.3f' % (name, calls, total, avg, pct))
