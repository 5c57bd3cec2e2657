"""Closest-hit rate of the raycast operator on incoherent rays: rays leave the first-hit points of the
frame in random directions of the hemisphere that faces the camera (what the second segments of the
paths look like).  Run under `rocprofv3 --kernel-trace --output-format csv` to read the kernels' own
durations (tools/raycast_rate.sh does, and lists them per call); the wall time printed here includes
the copies of the host arrays the operator takes.

    python tools/raycast_rate.py config/room.conf [--rays 4194304] [--variants "raycast_wavefront=1,wf_refill=24;..."]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("conf")
    ap.add_argument("--rays", type=int, default=1 << 22)
    ap.add_argument("--variants", default="", help="option sets (name=value,...) separated by ';': one context and two calls each")
    ap.add_argument("--sort", default="none", choices=["none", "origin", "origin+octant", "octant+origin"],
                    help="order of the rays: as drawn, by the Morton code of the origin's cell (2^bits per axis), the direction "
                         "octant within a cell, or the cell within an octant")
    ap.add_argument("--bits", type=int, default=7)
    a = ap.parse_args()
    s = pt.Scene.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, width=1024, height=1024, spp=1)
    c = pt.Context(0)
    c.build_accel(s)
    ids, pts = c.first_hit(with_points=True)
    pts = pts.reshape(-1, 3)[ids.reshape(-1) >= 0]
    rng = np.random.default_rng(13)
    pick = rng.integers(0, len(pts), a.rays)
    o = pts[pick]
    cam = np.asarray(s.camera()[1], dtype=np.float64)[:3]
    d = rng.normal(size=(a.rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = np.einsum("ij,ij->i", d, cam - o) < 0
    d[flip] *= -1
    if a.sort != "none":
        lo, hi = o.min(0), o.max(0)
        q = np.minimum(((o - lo) / (hi - lo + 1e-12) * (1 << a.bits)).astype(np.uint64), (1 << a.bits) - 1)
        morton = np.zeros(len(o), dtype=np.uint64)
        for b in range(a.bits):
            for k in range(3):
                morton |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + k)
        octant = ((d[:, 0] < 0).astype(np.uint64) | ((d[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((d[:, 2] < 0).astype(np.uint64) << np.uint64(2)))
        key = {"origin": morton, "origin+octant": (morton << np.uint64(3)) | octant, "octant+origin": (octant << np.uint64(3 * a.bits)) | morton}[a.sort]
        order = np.argsort(key, kind="stable")
        o, d = np.ascontiguousarray(o[order]), np.ascontiguousarray(d[order])
        print("rays sorted by %s, %d bits per axis" % (a.sort, a.bits), flush=True)
    ref = None
    for rep in range(3):
        t0 = time.time()
        hit, hp = c.raycast(o, d)
        dt = time.time() - t0
        ref = (hit, hp)
        print("operator: %d rays, %.1f ms wall (with copies), %.0f%% hit" % (a.rays, dt * 1e3, 100.0 * (hit >= 0).mean()), flush=True)
    for var in filter(None, a.variants.split(";")):
        cv = pt.Context(0)
        for kv in var.split(","):
            k, v = kv.split("=")
            cv.set_option(k, int(v))
        cv.build_accel(s)
        for rep in range(2):
            t0 = time.time()
            hit, hp = cv.raycast(o, d)
            dt = time.time() - t0
        same = int((hit != ref[0]).sum())
        far = float(np.nanmax(np.abs(hp - ref[1]))) if (hit >= 0).any() else 0.0
        print("%s: %.1f ms wall; %d face ids differ from the operator's, hit points within %.3g" % (var, dt * 1e3, same, far), flush=True)


if __name__ == "__main__":
    main()
