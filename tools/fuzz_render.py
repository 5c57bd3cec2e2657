"""Randomised image parity against the CPU oracle (needs a GPU): the Cornell box with random
geometry thrown in (free triangles, tilted and axis-aligned quads, slivers, fans; random materials
out of the box's own -- diffuse, glossy, glass, emitter -- flat or interpolated normals) and random
extra lights (spheres, a second square).  f64 renders must equal the oracle's at 1e-9 per pixel, every
exception matched to a named tie alternative of its path (the rule of oracle/parity.py), for both estimators,
with the BVH and -- up to 64 faces -- with the all-triangles intersector.  A scene whose random quads
coincide with a wall trips the guards of that rule (many tie pixels in a 24 x 24 frame): it is then
put through the proof sample by sample (prove_all) and fails only on a sample that equals none of
the values its path can take.

    python tools/fuzz_render.py [--seeds 30] [--first 0]
"""
import argparse
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402
from test_gpu_parity import assert_f64_parity  # noqa: E402


def extra_faces(rng, n):
    tris = []
    while len(tris) < n:
        kind = rng.integers(0, 5)
        c = rng.uniform(0.4, 5.1, 3)
        size = 10.0 ** rng.uniform(-1.5, 0.2)
        if kind == 0:
            tris.append(c + rng.normal(size=(3, 3)) * size)
        elif kind == 1:
            ax = rng.integers(0, 3)
            u, v = np.eye(3)[(ax + 1) % 3] * size, np.eye(3)[(ax + 2) % 3] * size * rng.uniform(0.3, 2)
            p = [c, c + u, c + u + v, c + v]
            tris += [np.array([p[0], p[1], p[2]]), np.array([p[0], p[2], p[3]])]
        elif kind == 2:
            u, v = rng.normal(size=3) * size, rng.normal(size=3) * size
            p = [c, c + u, c + u + v, c + v]
            tris += [np.array([p[0], p[1], p[2]]), np.array([p[0], p[2], p[3]])]
        elif kind == 3:
            a = c + rng.normal(size=3) * size
            tris.append(np.array([c, a, c + (a - c) * rng.uniform(0, 1) + rng.normal(size=3) * size * 1e-3]))
        else:
            k = int(rng.integers(3, 9))
            ring = [c + rng.normal(size=3) * size for _ in range(k)]
            tris += [np.array([c, ring[i], ring[(i + 1) % k]]) for i in range(k)]
    t = np.clip(np.asarray(tris[:n]), 0.05, 5.45)
    return t.astype(np.float32).astype(np.float64)


def scene(seed, template):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(0, 30), rng.integers(40, 200), rng.integers(200, 1200)]))
    ref = copy.deepcopy(template)
    if n:
        t = extra_faces(rng, n)
        nrm = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
        ln = np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-300), [0.0, 1.0, 0.0])
        vn = np.repeat(nrm[:, None, :], 3, axis=1)
        bend = rng.uniform(size=n) < 0.4                      # interpolated normals on some faces
        vn[bend] += rng.normal(size=(int(bend.sum()), 3, 3)) * 0.3
        vn /= np.linalg.norm(vn, axis=2, keepdims=True)
        ref.verts = np.concatenate([ref.verts, t])
        ref.normals = np.concatenate([ref.normals, vn.astype(np.float32).astype(np.float64)])
        ref.mat_ids = np.concatenate([ref.mat_ids, rng.choice([0, 2, 3, 3, 3, 4, 4, 5, 1], n)]).astype(np.int32)
    lights = [ref.lights[0]]
    for _ in range(int(rng.integers(0, 3))):
        L = np.zeros(16)
        L[0] = 1
        L[1:4] = rng.uniform(5, 60, 3)
        L[4:7] = rng.uniform(0.8, 4.8, 3)
        L[7] = rng.uniform(0.03, 0.3)
        lights.append(L)
    if rng.uniform() < 0.3:
        L = ref.lights[0].copy()
        L[4:16] = (L[4:16].reshape(4, 3) + [rng.uniform(-1.5, 1.5), -rng.uniform(0.5, 3), rng.uniform(-1.5, 1.5)]).reshape(-1)
        lights.insert(int(rng.integers(0, len(lights) + 1)), L)
    if rng.uniform() < 0.15:
        lights = lights[1:]                                  # sphere lights only / no light at all
    ref.lights = np.asarray(lights).reshape(-1, 16)
    ref.depth = int(rng.integers(1, 7))
    ref.spp = int(rng.integers(1, 4))
    return ref


def prove_all(c, O, ref, est, img, oimg, st, ost, seed=13, verbose=False):
    """The proof of oracle/parity.py without the test helper's guards (number of differing pixels,
    rMSE of the frame, segment counts), for scenes full of coincident faces: every sample of every
    pixel that differs -- of every pixel at all when only the segment counts differ -- is rendered
    alone and matched against the oracle's path or a named tie alternative of it.  Returns (samples
    that equal no alternative, samples that equal one, samples with a tie on their path that agree
    all the same)."""
    from oracle import parity
    diff = np.abs(img - oimg).max(axis=-1)
    everywhere = int(st["segments"]) != int(ost["segments"])
    pixels = [(y, x) for y in range(ref.height) for x in range(ref.width) if everywhere or diff[y, x] > parity.F64_TOL]
    r = parity.prove_pixels(c, O, pixels, range(ref.spp), seed=seed, render_kw=dict(estimator=est),
                            verbose=(lambda m: print("  " + m, flush=True)) if verbose else None)
    return len(r["unmatched"]), r["tie_samples"], r["silent_ties"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=30)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--size", type=int, default=24)
    a = ap.parse_args()
    template = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=a.size, height=a.size, spp=2)
    failures = 0
    kernels = {}   # tracer kernel -> renders it ran
    for seed in range(a.first, a.first + a.seeds):
        ref = scene(seed, template)
        s = pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights, ref.view,
                                 ref.cam_pos, ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp,
                                 ref.enable_indirect, ref.enable_bary_normal)
        notes = []
        for est in ("reference", "fast"):
            O = oracle.Oracle(ref, seed=13, estimator=est)
            oimg, ost = O.render(stats=True)
            for limit in ((64, 0) if ref.verts.shape[0] <= 64 else (0,)):
                c = pt.Context(0, brute_force_limit=limit)
                c.build_accel(s)
                img, st = c.render(mode="f64", seed=13, estimator=est)
                kernels[c.last_kernel()] = kernels.get(c.last_kernel(), 0) + 1
                try:
                    n_bad = assert_f64_parity(img, oimg, st, ost, depth=ref.depth, ctx=c, O=O, render_kw=dict(estimator=est))
                    if n_bad:
                        notes.append("%s/%s: %d tie pixels" % (est, c.last_kernel(), n_bad))
                except AssertionError as e:
                    # a guard of the test helper (count of differing pixels, rMSE of a 24 x 24 frame,
                    # segment counts with no pixel differing) or a sample without a tie: the proof
                    # without the guards tells which
                    un, ex, si = prove_all(c, O, ref, est, img, oimg, st, ost)
                    if un == 0 and (ex > 0 or si > 0):
                        notes.append("%s/%s: beyond the helper's guards, all ties (%d samples equal a named tie alternative, %d more meet a tie and agree)" % (
                            est, c.last_kernel(), ex, si))
                    else:
                        failures += 1
                        notes.append("%s/%s FAILED: %s (%d samples equal no alternative)" % (est, c.last_kernel(), str(e)[:300], un))
        print("seed %d: %d faces, %d lights, depth %d, %d spp%s" % (seed, ref.verts.shape[0], ref.lights.shape[0], ref.depth, ref.spp,
                                                                   "".join("; " + x for x in notes)), flush=True)
    for k in sorted(kernels):
        print("renders through %s: %d" % (k, kernels[k]), flush=True)
    print("TOTAL failures: %d" % failures, flush=True)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
