#!/usr/bin/env python3
"""Sweeps one context option on a scene (GPU box): Mpaths/s of a render per value, both precisions.
usage: sweep_option.py <conf> <option> <v1,v2,...> [size] [spp]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pathtracer_amd as pt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
conf, opt, values = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
size = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 32
scene = pt.Scene.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=size, height=size, spp=spp)
out = torch.zeros((size, size, 3), dtype=torch.float64, device="cuda")
for v in values:
    ctx = pt.Context(0)
    ctx.set_option(opt, v)
    ctx.build_accel(scene)
    row = []
    for mode in ("f64", "f32"):
        ctx.render_device(out.data_ptr(), mode=mode, spp=4)
        st = ctx.render_device(out.data_ptr(), mode=mode)
        row.append(st["paths"] / st["ms_total"] / 1e3)
    print("%s = %d: f64 %.1f  f32 %.1f Mpaths/s  (%s)" % (opt, v, row[0], row[1], ctx.last_kernel()), flush=True)
    del ctx
