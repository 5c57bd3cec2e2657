#!/usr/bin/env python3
"""Times one scene on the GPU: time_scene.py <conf> [width] [spp] [mode].  Prints Mpaths/s and statistics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
conf = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
mode = sys.argv[4] if len(sys.argv) > 4 else "f64"
scene = pt.Scene.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=W, spp=spp)
ctx = pt.Context(0)
for kv in filter(None, os.environ.get('PTX_OPTS', '').split(',')):
    k, v = kv.split('=')
    ctx.set_option(k, int(v))
ctx.build_accel(scene)
print("accel", ctx.accel_info(), "tris", scene.info.n_tris, "lights", scene.info.n_lights, flush=True)
out = torch.zeros((W, W, 3), dtype=torch.float64, device="cuda")
ctx.render_device(out.data_ptr(), mode=mode, spp=1)
st = ctx.render_device(out.data_ptr(), mode=mode, spp=spp)
util = st["segments"] / (64.0 * max(st["wave_iterations"], 1))
print("%s %dx%d %d spp %s: %.1f Mpaths/s (%.1f ms), %.2f segments/path, %.2f shadow rays/path, lane util %.3f" % (
    conf, W, W, spp, mode, st["paths"] / st["ms_total"] / 1e3, st["ms_total"], st["segments"] / st["paths"],
    st["shadow_rays"] / st["paths"], util), flush=True)
if st["bvh_nodes"]:
    rays = st["segments"] + st["shadow_rays"]
    print("BVH records visited per ray: %.2f nodes, %.2f triangles" % (st["bvh_nodes"] / rays, st["bvh_tris"] / rays), flush=True)
