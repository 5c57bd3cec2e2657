#!/usr/bin/env python3
"""Times a BVH scene WITH a square light (room.obj, its lamps replaced by one square light under the
ceiling) with the wavefront tracer, the ring tracer and the lane tracer: time_square_bvh.py [width] [spp] [mode]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pathtracer_amd as pt
from oracle import scene_io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "f64"
ref = scene_io.load_conf(os.path.join(ROOT, "config/room.conf"), root_dir=ROOT, width=W, height=W, spp=spp)
lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
c = 0.5 * (lo + hi)
e = 0.15 * (hi - lo)
y = hi[1] - 0.02 * (hi[1] - lo[1])
L = np.zeros(16)
L[0] = 0.0
L[1:4] = 20.0
L[4:16] = np.array([[c[0] - e[0], y, c[2] - e[2]], [c[0] + e[0], y, c[2] - e[2]], [c[0] + e[0], y, c[2] + e[2]], [c[0] - e[0], y, c[2] + e[2]]]).ravel()
ref.lights = np.array([L])
scene = pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights,
                             ref.view, ref.cam_pos, ref.fov_deg, ref.near, W, W, ref.depth, spp, ref.enable_indirect, ref.enable_bary_normal)
imgs = []
for rings, wavefront in ((1, 1), (1, 0), (0, 0)):   # wavefront tracer, ring tracer, lane tracer
    ctx = pt.Context(0)
    ctx.set_option("bvh_rings", rings)
    ctx.set_option("bvh_wavefront", wavefront)
    ctx.build_accel(scene)
    out = torch.zeros((W, W, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(out.data_ptr(), mode=mode, spp=1)
    st = ctx.render_device(out.data_ptr(), mode=mode, spp=spp)
    imgs.append(out.cpu().numpy())
    print("%s %s: %.1f Mpaths/s (%.1f ms), %.2f segments/path, %.2f shadow rays/path" % (
        ctx.last_kernel(), mode, st["paths"] / st["ms_total"] / 1e3, st["ms_total"], st["segments"] / st["paths"], st["shadow_rays"] / st["paths"]), flush=True)
print("identical:", bool(np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[1], imgs[2])))
