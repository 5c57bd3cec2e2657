"""The headline configuration at its full size against the CPU oracle: config/cbox.conf at
1024 x 1024 with 256 samples per pixel (2.7e8 paths) rendered in f64 on the GPU and by the oracle
(all host threads), compared pixel by pixel.  Pixels beyond 1e-9 are counted, and a sample of them is
put through the proof the parity tests use (oracle/parity.py): every sample of the pixel is rendered
alone on the GPU and re-traced by the oracle, and where the two differ the GPU's value must equal the
oracle's with the closest-hit ties of that path (two best candidates within 1e-10) falling another,
named way.  Prints one JSON line.

    python tools/full_frame_parity.py [--spp 256] [--size 1024] [--prove 40] [--conf config/mis.conf --width 768 --height 512]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # before the library: both bring a HIP runtime, and torch wants its own loaded first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402

TOL, TIE_GAP = 1e-9, 1e-10


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--conf", default="config/cbox.conf", help="another scene file: the same check on it")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--prove", type=int, default=40)
    a = ap.parse_args()
    W, H = a.width or a.size, a.height or a.size
    kw = dict(width=W, height=H, spp=a.spp)
    s = pt.Scene.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, **kw)
    ref = scene_io.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, **kw)
    if ref.verts.shape[0] > 256:
        ref.use_bvh = True                                 # the oracle's own culling tree; same hits as its plain loop
    c = pt.Context(0)
    c.build_accel(s)
    t0 = time.time()
    img, st = c.render(mode="f64", seed=13)
    t_gpu = time.time() - t0
    print("GPU frame done in %.1f s" % t_gpu, file=sys.stderr, flush=True)
    O = oracle.Oracle(ref, seed=13)
    t0 = time.time()
    oimg = np.zeros_like(img)
    ost = {"paths": 0, "segments": 0}
    rows = 64
    for r0 in range(0, H, rows):                           # in row bands, to show progress
        ref.region = (r0, 0, min(r0 + rows, H) - 1, W - 1)
        Ob = oracle.Oracle(ref, seed=13)
        part, pst = Ob.render(stats=True)
        oimg[r0:r0 + rows] = part[r0:r0 + rows]
        for k in ost:
            ost[k] += int(pst[k])
        print("oracle rows %d.. done (%.0f s)" % (r0, time.time() - t0), file=sys.stderr, flush=True)
    t_cpu = time.time() - t0
    diff = np.abs(img - oimg).max(axis=-1)
    bad = np.argwhere(diff > TOL)
    rec = {"config": "%s at %d x %d, %d spp, depth %d, f64" % (a.conf, W, H, a.spp, ref.depth), "kernel": c.last_kernel(),
           "paths": int(st["paths"]), "gpu_seconds": round(t_gpu, 2), "oracle_seconds": round(t_cpu, 1),
           "oracle_threads": int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count(),
           "segments_gpu": int(st["segments"]), "segments_oracle": ost["segments"],
           "pixels": int(diff.size), "pixels_beyond_1e-9": int(len(bad)), "largest_difference": float(diff.max()),
           "rmse_rel": float(oracle.rmse_rel(img, oimg))}
    ref.region = (0, 0, H - 1, W - 1)
    O = oracle.Oracle(ref, seed=13)
    from oracle import parity
    pick = bad[np.random.default_rng(13).permutation(len(bad))[: a.prove]] if len(bad) else np.zeros((0, 2), dtype=np.int64)
    proven = unexplained = 0
    if len(pick):
        dev = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")

        def render_row(y, smp):
            # the row's pixels, this sample alone (one launch, one row copied back)
            c.render_device(dev.data_ptr(), mode="f64", spp=1, sample_offset=smp, spp_total=1, rows=(y, y + 1), seed=13)
            return dev[y].cpu().numpy()

        r = parity.prove_pixels(c, O, [(int(y), int(x)) for y, x in pick], range(a.spp), seed=13, render_row=render_row)
        proven = r["proven_ties"]
        unexplained = len(pick) - proven - r["amplified"]
        for y, x, smp, detail in r["unmatched"][:20]:
            print("pixel (%d, %d) sample %d: NOT MATCHED %s" % (y, x, smp, detail), file=sys.stderr, flush=True)
        rec.update(tie_samples=r["tie_samples"], unmatched_samples=len(r["unmatched"]), pixels_no_sample_differs=r["no_sample_differs"],
                   amplified_roundings=r["amplified"], amplifications=[round(v) for v in r["amplifications"]])
    rec.update(pixels_put_to_the_proof=int(len(pick)), proven_ties=proven, not_explained=unexplained)
    print(json.dumps(rec), flush=True)
    return 0 if unexplained == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
