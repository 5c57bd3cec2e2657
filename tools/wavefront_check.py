"""Wavefront tracer against the ring tracer on one scene: largest difference of the f64 sums, statistics,
and the throughput of both.  python tools/wavefront_check.py config/room.conf [width] [spp] [mode] [estimator]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pathtracer_amd as pt
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
conf = sys.argv[1]
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 32
mode = sys.argv[4] if len(sys.argv) > 4 else "f64"
est = sys.argv[5] if len(sys.argv) > 5 else "reference"
scene = pt.Scene.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=W, height=W, spp=spp)
imgs = {}
for wf in (0, 1):
    ctx = pt.Context(0)
    ctx.set_option("bvh_wavefront", wf)
    for kv in filter(None, os.environ.get('PTX_OPTS', '').split(',')):
        k, v = kv.split('=')
        ctx.set_option(k, int(v))
    ctx.build_accel(scene)
    out = torch.zeros((W, W, 3), dtype=torch.float64, device="cuda")
    ctx.render_device(out.data_ptr(), mode=mode, spp=1, estimator=est)
    out.zero_()
    st = ctx.render_device(out.data_ptr(), mode=mode, spp=spp, estimator=est, **({"depth": int(os.environ["PTX_DEPTH"])} if os.environ.get("PTX_DEPTH") else {}))
    imgs[wf] = out.cpu().numpy()
    print("%s wavefront=%d %s: %.1f Mpaths/s (%.1f ms) kernel %s; segments %d tests %d records %d nodes %d tris %d iterations %d state %.0f MB" % (
        conf, wf, mode, st["paths"] / st["ms_total"] / 1e3, st["ms_total"], ctx.last_kernel(), st["segments"], st["shadow_rays"], st["records"],
        st["bvh_nodes"], st["bvh_tris"], st["wave_iterations"], st["state_bytes"] / 1e6), flush=True)
d = np.abs(imgs[0] - imgs[1])
bad = np.argwhere(d.max(axis=2) > 0)
for y, x in bad[:6]:
    print("  pixel (%d, %d): ring %s wavefront %s" % (x, y, imgs[0][y, x], imgs[1][y, x]))
print("largest difference of the sums: %.3g (pixels that differ at all: %d of %d)" % (d.max(), int((d.max(axis=2) > 0).sum()), W * W))
