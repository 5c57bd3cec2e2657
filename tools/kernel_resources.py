#!/usr/bin/env python3
"""Vector registers, LDS and scratch of every kernel of the library, from its gfx950 assembly
(hipcc ... --cuda-device-only -S context.hip -o context.s with the Makefile's flags), and the waves per
SIMD the registers allow (512 per lane and SIMD).   python tools/kernel_resources.py context.s"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        body = m.group(2)
        get = lambda k: int(re.search(k + r" (\d+)", body).group(1))
        rows.append((m.group(1), get(r"\.amdhsa_next_free_vgpr"), get(r"\.amdhsa_group_segment_fixed_size"), get(r"\.amdhsa_private_segment_fixed_size")))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
    for (_, vgpr, lds, scratch), name in zip(rows, names):
        name = name.split("(")[0].replace("ptx::", "").replace("void ", "")
        print("%-56s vgpr %3d  lds %6d  scratch %4d  waves per SIMD by registers %d" % (name[:56], vgpr, lds, scratch, min(8, 512 // max(vgpr, 1))))


if __name__ == "__main__":
    main()
