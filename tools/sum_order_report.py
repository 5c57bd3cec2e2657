#!/usr/bin/env python3
"""How much does the unpinned order of Eigen's short sums matter?

The reference's intersector leaves the order of its 3- and 4-term additions (dot products, the
determinant, row x vector, squared norms) to Eigen, whose source is not in the reference's checkout.
The CPU restatement is built in the three plausible orders (ORC_SUM 0 / 1 / 2 in oracle/pt_oracle.c)
and this script counts, per scene, what changes between them: first-hit face ids and hit points of
all camera rays, and pixels of a low-spp f64 image beyond the 1e-9 parity bar -- and for every such
pixel whether the path of one of its samples meets a closest-hit tie (winner and runner-up within
1e-10), i.e. whether the order only decides what the reference itself leaves to the last bit.

    python tools/sum_order_report.py [--size 1024] [--img 256] [--spp 4] > profiles/r02_sum_order_report.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle, scene_io  # noqa: E402

TIE_GAP = 1e-10


def run(conf, size, img, spp):
    out = {}
    ids, pts, ims, orcs = {}, {}, {}, {}
    for v in (0, 1, 2):
        oracle.use_sum_order(v)
        ref = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=size, height=size, spp=1)
        O = oracle.Oracle(ref, seed=13)
        ids[v], pts[v] = O.first_hit(with_points=True)
        small = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=img, height=img, spp=spp)
        Os = oracle.Oracle(small, seed=13)
        ims[v] = Os.render()
        orcs[v] = (Os, small)          # keeps the scene arrays alive; only valid while variant v is loaded
    out["camera_rays"] = int(ids[0].size)
    out["image"] = "%dx%d, %d spp" % (img, img, spp)
    for v in (1, 2):
        same_pts = ((pts[v] == pts[0]) | (np.isnan(pts[v]) & np.isnan(pts[0]))).all(axis=1)
        diff = np.abs(ims[v] - ims[0]).max(axis=-1)
        bad = np.argwhere(diff > 1e-9)
        # a tie on the path of one of the pixel's samples, as either order traces it; otherwise:
        # do both orders walk the same faces (then only the arithmetic along the path differs)?
        tied = np.zeros(len(bad), bool)
        faces = {}
        for order in (0, v):
            oracle.use_sum_order(order)
            small = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=img, height=img, spp=spp)
            Oo = oracle.Oracle(small, seed=13)
            for k, (r, c) in enumerate(bad):
                tr = [Oo.trace_sample(int(c), int(r), s) for s in range(spp)]
                faces[(order, k)] = [t[1].tolist() for t in tr]
                if any(t[2].min(initial=np.inf) <= TIE_GAP for t in tr):
                    tied[k] = True
        ties = int(tied.sum())
        same_path = sum(1 for k in range(len(bad)) if not tied[k] and faces[(0, k)] == faces[(v, k)])
        untied_max = max([float(diff[r, c]) for k, (r, c) in enumerate(bad) if not tied[k]], default=0.0)
        out["order_%d_vs_0" % v] = {
            "first_hit_ids_changed": int((ids[v] != ids[0]).sum()),
            "hit_points_changed_in_some_bit": int((~same_pts).sum()),
            "largest_hit_point_change": float(np.nanmax(np.abs(pts[v] - pts[0]))) if (~same_pts).any() else 0.0,
            "image_pixels_beyond_1e-9": int(len(bad)),
            "of_which_with_a_closest_hit_tie_on_a_path": ties,
            "of_the_rest_on_unchanged_paths": same_path,
            "largest_change_without_a_tie": untied_max,
            "largest_image_change": float(diff.max()),
        }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--scenes", default="cbox,mis,room")
    a = ap.parse_args()
    rep = {"what": __doc__.split("\n\n")[0] + " Orders: 0 (a+b)+c, ((a+b)+c)+d; 1 a+(b+c), (a+b)+(c+d); 2 a+(b+c), (a+c)+(b+d).",
           "tie_gap": TIE_GAP, "scenes": {}}
    for name in a.scenes.split(","):
        t0 = time.time()
        rep["scenes"][name] = run("config/%s.conf" % name, a.size, a.img, a.spp)
        rep["scenes"][name]["seconds"] = round(time.time() - t0, 1)
    oracle.use_sum_order(0)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
