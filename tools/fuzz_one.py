"""Single seeds of tools/fuzz_render.py again, with the whole assertion text and the differing pixels.

    python tools/fuzz_one.py 2045 2079 [--size 24]
"""
import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402
from test_gpu_parity import assert_f64_parity  # noqa: E402
from fuzz_render import scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seeds", type=int, nargs="+")
    ap.add_argument("--size", type=int, default=24)
    a = ap.parse_args()
    template = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=a.size, height=a.size, spp=2)
    bad = 0
    for seed in a.seeds:
        ref = scene(seed, template)
        s = pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights, ref.view,
                                 ref.cam_pos, ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp,
                                 ref.enable_indirect, ref.enable_bary_normal)
        for est in ("reference", "fast"):
            O = oracle.Oracle(ref, seed=13, estimator=est)
            oimg, ost = O.render(stats=True)
            for opts in ({}, {"bvh_wavefront": 0}, {"bvh_wavefront": 0, "bvh_rings": 0}):
                c = pt.Context(0, brute_force_limit=0)
                for k, v in opts.items():
                    c.set_option(k, v)
                c.build_accel(s)
                img, st = c.render(mode="f64", seed=13, estimator=est)
                d = np.abs(img - oimg).max(axis=2)
                print("seed %d %s %s %s: %d pixels beyond 1e-9 (largest %.3g); segments %d / oracle %d, tests %d / %d" % (
                    seed, est, opts, c.last_kernel(), int((d > 1e-9).sum()), d.max(), st["segments"], ost["segments"], st["shadow_rays"], ost["shadow_rays"]), flush=True)
                try:
                    assert_f64_parity(img, oimg, st, ost, depth=ref.depth, ctx=c, O=O, render_kw=dict(estimator=est))
                except AssertionError:
                    bad += 1
                    traceback.print_exc(limit=3, file=sys.stdout)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
