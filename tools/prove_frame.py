#!/usr/bin/env python3
"""Every f64 outlier of one frame against the oracle, sample by sample, with the details of whatever
stays unmatched (GPU box).

    python tools/prove_frame.py config/room.conf --width 3840 --height 2160 --spp 1 [--max-pixels 200]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the library: torch wants its own HIP runtime loaded first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, parity, scene_io  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("conf")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--max-pixels", type=int, default=0)
    ap.add_argument("--gap", type=float, default=parity.TIE_GAP)
    a = ap.parse_args()
    kw = dict(width=a.width, height=a.height, spp=a.spp)
    s = pt.Scene.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, **kw)
    ref = scene_io.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, **kw)
    ctx = pt.Context(0)
    ctx.build_accel(s)
    O = oracle.Oracle(ref, seed=13)
    gimg, _ = ctx.render(mode="f64", spp=a.spp, seed=13)
    oimg = O.render(spp=a.spp, threads=int(os.environ.get("OMP_NUM_THREADS", "16")))
    bad, diff = parity.outliers(gimg, oimg)
    print("pixels beyond 1e-9: %d, largest difference %.3g" % (len(bad), diff.max()), flush=True)
    if a.max_pixels and len(bad) > a.max_pixels:
        bad = bad[:a.max_pixels]
    r = parity.prove_pixels(ctx, O, [(int(y), int(x)) for y, x in bad], range(a.spp), seed=13, gap=a.gap)
    un = r.pop("unmatched")
    r["amplifications"] = [round(v) for v in r["amplifications"]]
    print(json.dumps(r))
    for y, x, smp, detail in un:
        rgb0, faces, margins, nt = O.trace_sample_ties(x, y, smp, 0, a.gap)
        one, _ = ctx.render(mode="f64", spp=1, sample_offset=smp, spp_total=1, rows=(y, y + 1), seed=13)
        print("UNMATCHED pixel (row %d, col %d) sample %d: gpu %s oracle %s ties %d faces %s margins %s nearest alternative %.3g" % (
            y, x, smp, one[y][x].tolist(), rgb0.tolist(), nt, faces.tolist(), ["%.3g" % m for m in margins.tolist()], detail.get("nearest_alternative", -1)), flush=True)


if __name__ == "__main__":
    main()
