#!/usr/bin/env python3
"""Quick GPU-vs-oracle diagnostic (prints numbers, asserts nothing).  Run on a GPU box."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
from oracle import scene_io, oracle

conf = sys.argv[1] if len(sys.argv) > 1 else "config/cbox.conf"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W
ref = scene_io.load_conf(os.path.join(root, conf), root_dir=root, width=W, height=H, spp=spp)
O = oracle.Oracle(ref)
s = pt.Scene.load_conf(os.path.join(root, conf), root_dir=root, width=W, height=H, spp=spp)
c = pt.Context(0)
c.build_accel(s)
print("accel", c.accel_info())
o, d = c.generate_rays()
oo, od = O.generate_rays()
print("rays bit-equal:", np.array_equal(o, oo), np.array_equal(d, od), np.abs(d - od).max())
ids, pts = c.first_hit(with_points=True)
oids, opts = O.first_hit(with_points=True)
print("first-hit mismatches:", int((ids != oids).sum()), "of", ids.size, "pts max diff", np.nanmax(np.abs(pts - opts)))
t = time.time(); img, st = c.render(mode="f64"); dt = time.time() - t
print("gpu render", dt, st)
t = time.time(); oimg, ost = O.render(stats=True); odt = time.time() - t
print("oracle render", odt, ost)
diff = np.abs(img - oimg)
print("max abs diff", diff.max(), "pixels differing >1e-9:", int((diff.max(axis=2) > 1e-9).sum()), "of", W * H,
      "rMSE", oracle.rmse_rel(img, oimg))
print("segments gpu/oracle", st["segments"], ost["segments"], "shadow", st["shadow_rays"], ost["shadow_rays"])
img32, st32 = c.render(mode="f32")
print("f32 render", st32["ms_total"], "rMSE vs oracle", oracle.rmse_rel(img32, oimg))
o2 = O.render(spp=spp)  # same seed -> identical
O2 = oracle.Oracle(ref, seed=14)
oimg2 = O2.render()
print("oracle seed noise floor rMSE", oracle.rmse_rel(oimg2, oimg))
