#!/usr/bin/env python3
"""mis.obj under twenty lights (six of them square) on the wavefront tracer and on the tracers it replaces
for such scenes, both estimators (GPU box).   python tools/time_many_lights.py [width height spp]"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import scene_io  # noqa: E402  (scene arrays only: nothing is rendered on the CPU here)

W, H, SPP = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (768, 512, 64)))
ref = scene_io.load_conf(os.path.join(ROOT, "config/mis.conf"), root_dir=ROOT, width=W, height=H, spp=SPP)
lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
rng = np.random.default_rng(11)
lights = list(ref.lights)
for k in range(15):
    L = np.zeros(ref.lights.shape[1])
    L[1:4] = rng.uniform(0.5, 4.0, 3)
    if k % 5 < 2:
        x, z = rng.uniform(lo[[0, 2]] + 0.1 * (hi - lo)[[0, 2]], hi[[0, 2]] - 0.2 * (hi - lo)[[0, 2]])
        e, y = 0.08 * (hi - lo)[[0, 2]], hi[1] - rng.uniform(0.02, 0.1) * (hi[1] - lo[1])
        L[4:16] = np.array([[x, y, z], [x + e[0], y, z], [x + e[0], y, z + e[1]], [x, y, z + e[1]]]).ravel()
    else:
        L[0] = 1.0
        L[4:7] = rng.uniform(lo, hi)
        L[7] = rng.uniform(0.02, 0.1) * float((hi - lo).max())
    lights.insert(int(rng.integers(0, len(lights) + 1)), L)
ref.lights = np.array(lights)
scene = pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights, ref.view, ref.cam_pos,
                             ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp, ref.enable_indirect, ref.enable_bary_normal)
out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
for wf in (1, 0):
    c = pt.Context(0)
    c.set_option("bvh_wavefront", wf)
    c.build_accel(scene)
    for est in ("reference", "fast"):
        for mode in ("f64", "f32"):
            c.render_device(out.data_ptr(), mode=mode, spp=1, estimator=est)
            st = c.render_device(out.data_ptr(), mode=mode, estimator=est)
            print("bvh_wavefront=%d %s %s: %.1f Mpaths/s, %.2f tests per path (%s)" % (wf, est, mode, st["paths"] / st["ms_total"] / 1e3,
                  st["shadow_rays"] / st["paths"], c.last_kernel()), flush=True)
