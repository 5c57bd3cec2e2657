"""Randomised parity check of the intersectors against the CPU oracle (needs a GPU): random triangle
soups with the shapes that stress a culling structure -- coplanar quads split along a diagonal,
exact duplicates, slivers, zero-area faces, fans round one vertex, far-from-origin clusters -- and
rays that stress the tests -- axis-parallel directions, origins on surfaces, rays aimed at vertices
and along edges and face planes.  ptx_raycast (ids and points) and ptx_vistest must equal the
oracle's plain loop over all faces bit for bit, with the all-triangles intersector (<= 64 faces) and
with the BVH.

One class of difference is expected and counted apart: a ray that lies IN the supporting plane of a
triangle makes the reference's 3 x 3 system singular, and what its inverse then returns is rounding
noise that can land inside the acceptance window at a point well away from the triangle.  The
all-triangles intersector runs the same arithmetic on every face and reproduces that; the BVH only
visits triangles whose box the ray meets.

    python tools/fuzz_parity.py [--seeds 40] [--first 0] [--rays 6000] [--max-faces 3000]
"""
import argparse
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402
from oracle import oracle, scene_io  # noqa: E402


def soup(rng, n):
    """n triangles [n,3,3], float32-representable."""
    scale = 10.0 ** rng.uniform(-1, 2)
    tris = []
    while len(tris) < n:
        kind = rng.integers(0, 8)
        c = rng.uniform(-scale, scale, 3)
        size = scale * 10.0 ** rng.uniform(-3, 0)
        if kind == 0:      # a free triangle
            tris.append(c + rng.normal(size=(3, 3)) * size)
        elif kind == 1:    # an axis-aligned quad, two triangles sharing the diagonal
            ax = rng.integers(0, 3)
            u, v = np.eye(3)[(ax + 1) % 3] * size, np.eye(3)[(ax + 2) % 3] * size * rng.uniform(0.2, 3)
            p = [c, c + u, c + u + v, c + v]
            tris += [np.array([p[0], p[1], p[2]]), np.array([p[0], p[2], p[3]])]
        elif kind == 2:    # a tilted quad
            u, v = rng.normal(size=3) * size, rng.normal(size=3) * size
            p = [c, c + u, c + u + v, c + v]
            tris += [np.array([p[0], p[1], p[2]]), np.array([p[0], p[2], p[3]])]
        elif kind == 3 and tris:    # an exact duplicate of an earlier face
            tris.append(tris[rng.integers(0, len(tris))].copy())
        elif kind == 4:    # a sliver
            a = c + rng.normal(size=3) * size
            tris.append(np.array([c, a, c + (a - c) * rng.uniform(0, 1) + rng.normal(size=3) * size * 1e-5]))
        elif kind == 5:    # no area
            a = c + rng.normal(size=3) * size
            tris.append(np.array([c, a, a]) if rng.integers(0, 2) else np.array([c, c, c]))
        elif kind == 6:    # a fan round one vertex
            k = int(rng.integers(3, 12))
            ring = [c + rng.normal(size=3) * size for _ in range(k)]
            tris += [np.array([c, ring[i], ring[(i + 1) % k]]) for i in range(k)]
        else:              # a small cluster far from the origin
            far = c + rng.choice([-1, 1], 3) * scale * 50
            tris += [far + rng.normal(size=(3, 3)) * size for _ in range(3)]
    return np.asarray(tris[:n], dtype=np.float32).astype(np.float64)


def rays(rng, verts, n):
    F = verts.shape[0]
    lo, hi = verts.reshape(-1, 3).min(0), verts.reshape(-1, 3).max(0)
    ext = np.maximum(hi - lo, 1e-3)
    o = rng.uniform(lo - 0.2 * ext, hi + 0.2 * ext, (n, 3))
    d = rng.normal(size=(n, 3))
    # a point on a random face
    w = rng.dirichlet([1, 1, 1], n)
    f = rng.integers(0, F, n)
    on = np.einsum("nk,nkc->nc", w, verts[f])
    kind = rng.integers(0, 8, n)
    for i in range(n):
        k = kind[i]
        if k == 1:      # towards a point on a face
            d[i] = on[i] - o[i]
        elif k == 2:    # towards a vertex
            d[i] = verts[f[i], rng.integers(0, 3)] - o[i]
        elif k == 3:    # towards a point on an edge
            t = rng.uniform()
            d[i] = verts[f[i], 0] * t + verts[f[i], 1] * (1 - t) - o[i]
        elif k == 4:    # axis-parallel
            d[i] = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1, 1])
            if rng.integers(0, 2):
                o[i] = on[i] - d[i] * ext.max() * rng.uniform(0.1, 2)
        elif k == 5:    # from a surface point, any direction
            o[i] = on[i]
        elif k == 6:    # from a surface point, inside the face's plane
            o[i] = on[i]
            d[i] = verts[f[i], 1] - verts[f[i], 0] + (verts[f[i], 2] - verts[f[i], 0]) * rng.uniform(-1, 1)
        elif k == 7:    # one zero component
            d[i, rng.integers(0, 3)] = 0.0
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    bad = (nrm[:, 0] == 0) | ~np.isfinite(nrm[:, 0])
    d[bad] = [0.0, 0.0, 1.0]
    nrm[bad] = 1.0
    return o, d / nrm, on


def one(seed, n_rays, template, max_faces=3000):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([rng.integers(1, 65), rng.integers(65, 400), rng.integers(400, max(max_faces, 401))]))
    ref = copy.deepcopy(template)
    ref.verts = soup(rng, n).reshape(-1, 3, 3)
    ref.normals = np.full_like(ref.verts, np.nan)
    ref.mat_ids = np.zeros(n, dtype=np.int32)
    s = pt.Scene.from_arrays(ref.verts.astype(np.float32), None, ref.mat_ids, ref.mats, ref.lights, ref.view, ref.cam_pos,
                             ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp)
    O = oracle.Oracle(ref, seed=13)
    o, d, on = rays(rng, ref.verts, n_rays)
    want = [O.closest_hit(o[i], d[i]) for i in range(n_rays)]
    w_ids = np.array([w[0] for w in want], dtype=np.int32)
    w_pts = np.array([w[1] if w[0] >= 0 else [np.nan] * 3 for w in want])
    a, b = on[: n_rays // 4], on[n_rays // 4: n_rays // 2]
    w_vis = np.array([O.vistest(a[i], b[i]) for i in range(a.shape[0])])
    bad = 0
    singular = [0]
    for limit in ((64, 0) if n <= 64 else (0,)):
        c = pt.Context(0, brute_force_limit=limit)
        c.build_accel(s)
        ids, pts = c.raycast(o, d)
        vis = c.vistest(a, b)
        wrong = (ids != w_ids) | ((pts != w_pts) & ~(np.isnan(pts) & np.isnan(w_pts))).any(1)
        if limit == 0:
            # rays in the plane of the face either side reports: the singular case
            for i in np.flatnonzero(wrong):
                for face in (w_ids[i], ids[i]):
                    if face >= 0:
                        t = ref.verts[face]
                        nrm = np.cross(t[1] - t[0], t[2] - t[0])
                        ln = np.linalg.norm(nrm)
                        if ln == 0 or abs(nrm @ d[i]) / ln < 1e-9:
                            wrong[i] = False
                            singular[0] += 1
        if wrong.any() or (vis != w_vis).any():
            bad += int(wrong.sum()) + int((vis != w_vis).sum())
            i = int(np.flatnonzero(wrong)[0]) if wrong.any() else -1
            print("seed %d, %d faces, %s: %d rays and %d segments differ; first ray %d: gpu %s %s oracle %s %s" % (
                seed, n, "BVH" if limit == 0 else "all-triangles", int(wrong.sum()), int((vis != w_vis).sum()), i,
                ids[i] if i >= 0 else "", pts[i] if i >= 0 else "", w_ids[i] if i >= 0 else "", w_pts[i] if i >= 0 else ""), flush=True)
    return n, bad, float((w_ids >= 0).mean()), float(w_vis.mean()), singular[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--rays", type=int, default=6000)
    ap.add_argument("--max-faces", type=int, default=3000)
    a = ap.parse_args()
    template = scene_io.load_conf(os.path.join(ROOT, "config/cbox.conf"), root_dir=ROOT, width=8, height=8, spp=1)
    total = sing = 0
    for seed in range(a.first, a.first + a.seeds):
        n, bad, hit, vis, s1 = one(seed, a.rays, template, a.max_faces)
        total += bad
        sing += s1
        print("seed %d: %d faces, %.0f%% of the rays hit, %.0f%% of the segments clear, %d differences%s" % (
            seed, n, 100 * hit, 100 * vis, bad, ", %d in-plane rays answered differently by the BVH" % s1 if s1 else ""), flush=True)
    print("TOTAL differences: %d (and %d in-plane rays, the singular case)" % (total, sing), flush=True)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
