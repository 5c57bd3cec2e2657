"""One seed of tools/fuzz_render.py looked at closely: every differing pixel is put through the tie
proof of tests/test_gpu_parity.py whatever their number (margins and faces printed), and the raycast
operator is checked against the oracle's closest hit on random rays through the scene.

    python tools/fuzz_prove.py 2079 [--rays 3000]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402
from fuzz_render import prove_all, scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seed", type=int)
    ap.add_argument("--size", type=int, default=24)
    ap.add_argument("--rays", type=int, default=3000)
    a = ap.parse_args()
    template = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=a.size, height=a.size, spp=2)
    ref = scene(a.seed, template)
    s = pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights, ref.view,
                             ref.cam_pos, ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp,
                             ref.enable_indirect, ref.enable_bary_normal)
    c = pt.Context(0, brute_force_limit=0)
    c.build_accel(s)
    unexplained = 0
    for est in ("reference", "fast"):
        O = oracle.Oracle(ref, seed=13, estimator=est)
        oimg, ost = O.render(stats=True)
        img, st = c.render(mode="f64", seed=13, estimator=est)
        diff = np.abs(img - oimg).max(axis=-1)
        print("%s: %d pixels differ; segments %d / %d" % (est, int((diff > 1e-9).sum()), st["segments"], ost["segments"]), flush=True)
        un, ex, si = prove_all(c, O, ref, est, img, oimg, st, ost, verbose=True)
        print("%s: %d samples differ on a proven tie, %d meet a tie and agree, %d differ without one" % (est, ex, si, un), flush=True)
        unexplained += un
    # the operator against the oracle's closest hit
    O = oracle.Oracle(ref, seed=13)
    rng = np.random.default_rng(a.seed)
    lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
    o = rng.uniform(lo, hi, (a.rays, 3))
    d = rng.normal(size=(a.rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ids, pts = c.raycast(o, d)
    wrong = 0
    for i in range(a.rays):
        f, p = O.closest_hit(o[i], d[i])
        if f != ids[i] or not np.array_equal(p, pts[i], equal_nan=True):
            wrong += 1
            print("  ray %d: operator face %d, oracle face %d" % (i, ids[i], f), flush=True)
    print("raycast operator: %d of %d random rays differ from the oracle's closest hit; %.0f%% hit" % (wrong, a.rays, 100.0 * (ids >= 0).mean()), flush=True)
    print("unexplained samples: %d" % unexplained, flush=True)
    return 1 if unexplained or wrong else 0


if __name__ == "__main__":
    sys.exit(main())
