#!/bin/bash
# the physically based estimator on room and mis, f64, at 1024 x 1024 x 32 spp (GPU box)
for conf in config/room.conf config/mis.conf; do
python3 - $conf <<'PY' || exit 1
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import pathtracer_amd as pt
conf = sys.argv[1]
scene = pt.Scene.load_conf(conf, root_dir=".", width=1024, height=1024, spp=32)
ctx = pt.Context(0)
for kv in filter(None, os.environ.get('PTX_OPTS', '').split(',')):
    k, v = kv.split('='); ctx.set_option(k, int(v))
ctx.build_accel(scene)
out = torch.zeros((1024, 1024, 3), dtype=torch.float64, device="cuda")
for mode in ("f64", "f32"):
    ctx.render_device(out.data_ptr(), mode=mode, spp=1, estimator="fast")
    st = ctx.render_device(out.data_ptr(), mode=mode, estimator="fast")
    print("%s fast %s: %.1f Mpaths/s (%.1f ms), %.2f segments and %.2f tests per path (%s)" % (
        conf, mode, st["paths"] / st["ms_total"] / 1e3, st["ms_total"], st["segments"] / st["paths"], st["shadow_rays"] / st["paths"], ctx.last_kernel()), flush=True)
PY
done
