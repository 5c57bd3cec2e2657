#!/usr/bin/env python3
"""Where along its path does one sample of one pixel part from the oracle?  The sample is rendered at
depth 1, 2, ... by the GPU (f64) and by the oracle; the first depth at which the two differ names the
bounce (GPU box).   python tools/depth_scan_pixel.py config/room.conf 3840 2160 567:210 884:1300 ..."""
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402

conf, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pix = [tuple(int(v) for v in a.split(":")) for a in sys.argv[4:]]
for depth in range(1, 6):
    s = pt.Scene.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=H, spp=1, depth=depth)
    ref = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=H, spp=1, depth=depth)
    ctx = pt.Context(0)
    ctx.build_accel(s)
    O = oracle.Oracle(ref, seed=13)
    for y, x in pix:
        one, _ = ctx.render(mode="f64", spp=1, sample_offset=0, spp_total=1, rows=(y, y + 1), seed=13)
        rgb = O.trace_sample(x, y, 0)
        rgb = rgb[0] if isinstance(rgb, tuple) else rgb
        print("depth %d pixel (%d, %d): gpu %s  oracle %s  diff %.3g" % (depth, y, x, one[y][x].tolist(), np.asarray(rgb).tolist(), float(np.abs(one[y][x] - np.asarray(rgb)).max())), flush=True)
