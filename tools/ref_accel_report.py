#!/usr/bin/env python3
"""How often do the reference's own culling structures (uniform grid "AABB", "OCTREE") answer
differently from its plain loop over all faces?   usage: ref_accel_report.py <conf> [width] [bounce rays]

Camera rays of the conf's camera plus random secondary rays that start on the first-hit surfaces are
cast through the plain loop (oracle/pt_oracle.c) and through the restated structures
(oracle/ref_accels.c).  Prints one JSON object.  CPU only."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle, scene_io  # noqa: E402


def rays_of(ref, n_bounce, seed=13):
    """Camera rays, and n_bounce rays leaving random first-hit points in random directions."""
    O = oracle.Oracle(ref, seed=seed)
    o, d = O.generate_rays()
    ids, pts = O.first_hit(with_points=True)
    rng = np.random.default_rng(seed)
    hit = np.flatnonzero(ids >= 0)
    pick = rng.choice(hit, size=min(n_bounce, hit.size), replace=False) if hit.size else hit
    d2 = rng.normal(size=(pick.size, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    return O, (o, d, ids, pts), (pts[pick], d2)


def compare(brute_ids, brute_pts, ids, pts, o):
    differ = ids != brute_ids
    db = np.linalg.norm(brute_pts - o, axis=1)
    da = np.linalg.norm(pts - o, axis=1)
    both = differ & (ids >= 0) & (brute_ids >= 0)
    return {"rays": int(ids.size), "different_face": int(differ.sum()),
            "hit_lost": int((differ & (ids < 0)).sum()),
            "farther_face": int((both & (da > db)).sum()),
            "tie_other_face": int((both & (da == db)).sum()),
            "closer_than_plain_loop": int(((ids >= 0) & (brute_ids >= 0) & (da < db)).sum() + ((ids >= 0) & (brute_ids < 0)).sum()),
            "same_face_other_point": int(((~differ) & (ids >= 0) & (pts != brute_pts).any(1)).sum())}


def report(conf, width=128, n_bounce=20000, structures=None):
    ref = scene_io.load_conf(os.path.join(ROOT, conf), root_dir=ROOT, width=width, height=width, spp=1)
    if ref.verts.shape[0] > 256:
        ref.use_bvh = True
    O, (o, d, ids, pts), (o2, d2) = rays_of(ref, n_bounce)
    ids2 = np.empty(o2.shape[0], dtype=np.int32)
    pts2 = np.empty_like(o2)
    for i in range(o2.shape[0]):
        ids2[i], pts2[i] = O.closest_hit(o2[i], d2[i])
    out = {"conf": conf, "faces": int(ref.verts.shape[0]), "structures": {}}
    for kind, param in structures or (("AABB", 2), ("AABB", 4), ("OCTREE", 30), ("OCTREE", 8)):
        t0 = time.perf_counter()
        try:
            A = oracle.RefAccel(ref.verts, kind, param)
        except RuntimeError as e:
            out["structures"]["%s %d" % (kind, param)] = {"error": str(e)}
            continue
        a_ids, a_pts = A.raycast(o, d)
        b_ids, b_pts = A.raycast(o2, d2)
        out["structures"]["%s %d" % (kind, param)] = dict(
            A.info(), build_and_query_s=round(time.perf_counter() - t0, 2),
            camera=compare(ids, pts, a_ids, a_pts, o), secondary=compare(ids2, pts2, b_ids, b_pts, o2))
        if kind == "OCTREE":
            # the same octree as the reference finds it on its next start: written to its text cache
            # (bounds at six significant digits) and read back
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, "mesh.obj.octree")
                A.write_octree(path)
                size = os.path.getsize(path)
                try:
                    B = oracle.RefAccel.from_octree_file(ref.verts, path)
                except RuntimeError as e:
                    out["structures"]["%s %d, read back from its cache" % (kind, param)] = {"error": str(e), "cache_bytes": size}
                    continue
            c_ids, c_pts = B.raycast(o, d)
            d_ids, d_pts = B.raycast(o2, d2)
            out["structures"]["%s %d, read back from its cache" % (kind, param)] = dict(
                B.info(), cache_bytes=size,
                camera=compare(ids, pts, c_ids, c_pts, o), secondary=compare(ids2, pts2, d_ids, d_pts, o2),
                camera_rays_answered_unlike_the_built_octree=int(((c_ids != a_ids) | ((c_pts != a_pts).any(1) & (c_ids >= 0))).sum()),
                secondary_rays_answered_unlike_the_built_octree=int(((d_ids != b_ids) | ((d_pts != b_pts).any(1) & (d_ids >= 0))).sum()))
    return out


if __name__ == "__main__":
    conf = sys.argv[1] if len(sys.argv) > 1 else "config/cbox.conf"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
    print(json.dumps(report(conf, w, nb), indent=1))
