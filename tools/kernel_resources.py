#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel: parses the remarks of
hipcc -Rpass-analysis=kernel-resource-usage (stderr of a compile) given as a file."""
import re, sys
cur = {}
rows = []
for line in open(sys.argv[1]):
    m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
for r in rows:
    if len(sys.argv) > 2 and sys.argv[2] not in r["name"]:
        continue
    print("%-70s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %2s lds %6s" % (r["name"][:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
