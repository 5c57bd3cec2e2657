#!/usr/bin/env python3
"""Sweeps batch size and region size on the headline workload (GPU box)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pathtracer_amd as pt
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "f64"
regions = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else "64,128,256,512,1024".split(","))]
batches = [int(x) for x in (sys.argv[4].split(",") if len(sys.argv) > 4 else "22,23,24".split(","))]
ppws = [int(x) for x in (sys.argv[5].split(",") if len(sys.argv) > 5 else ["0"])]
scene = pt.Scene.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=1024, height=1024, spp=spp)
out = torch.zeros((1024, 1024, 3), dtype=torch.float64, device="cuda")
for lb, reg, ppw in itertools.product(batches, regions, ppws):
    ctx = pt.Context(0, batch_paths=1 << lb, chunk_paths=ppw, trace_blocks_per_cu=int(os.environ.get('PTX_BPC', '0')))
    for kv in filter(None, os.environ.get('PTX_OPTS', '').split(',')):
        k, v = kv.split('=')
        pt.api.check(pt._lib.lib().ptx_context_set_option(ctx._h, k.encode(), int(v)))
    ctx.build_accel(scene)
    ctx.render_device(out.data_ptr(), mode=mode, spp=8)
    st = ctx.render_device(out.data_ptr(), mode=mode, spp=spp)
    sp = ctx.render_device(out.data_ptr(), mode=mode, spp=spp, profile=True)
    util = st["segments"] / (64.0 * max(st["wave_iterations"], 1))
    print("batch 2^%d region %4d chunk %5d: %8.1f Mpaths/s  ms %.1f | profiled: trace %.1f accumulate %.1f other %.1f | lane util %.3f" % (
        lb, reg, ppw, st["paths"] / st["ms_total"] / 1e3, st["ms_total"], sp["ms_trace"], sp["ms_accumulate"], sp["ms_other"], util), flush=True)
    del ctx
