"""CPU tests of the host-side acceleration data of the all-triangles intersector (plane records,
coplanar pairing, blocker sets), read back through ptx_scene_plane_records and checked against the
oracle's exact ray/triangle test."""
import os

import numpy as np

import pathtracer_amd as pt
from oracle import oracle, scene_io


def load(root, conf, **kw):
    s = pt.Scene.load_conf(os.path.join(root, conf), root_dir=root, **kw)
    ref = scene_io.load_conf(os.path.join(root, conf), root_dir=root, **kw)
    return s, ref


def test_cornell_box_quads_pair_up_exactly(root):
    s, ref = load(root, "config/cbox.conf", width=8, height=8, spp=1)
    n_recs, n_pairs, rec_tri, blockers = s.plane_records()
    # 13 quad faces; the red wall's four corners are not coplanar, so its halves stay single records
    assert (n_recs, n_pairs) == (14, 12)
    used = rec_tri[rec_tri >= 0]
    assert sorted(used.tolist()) == list(range(26))             # every triangle exactly once
    verts = s.geometry()[0].reshape(26, 3, 3).astype(np.float64)
    for r in range(n_recs):
        a, b = rec_tri[2 * r], rec_tri[2 * r + 1]
        assert a >= 0
        if b >= 0:                                              # a pair: same fan-triangulated quad, coplanar
            assert a // 2 == b // 2
            n = np.cross(verts[a][1] - verts[a][0], verts[a][2] - verts[a][0])
            assert np.abs((verts[b] - verts[a][0]) @ n).max() == 0.0
    assert (rec_tri[2 * n_recs:] == -1).all()


def test_blocker_sets_cover_every_real_blocker(root):
    """No triangle outside the blocker set of a record ever blocks a segment from that record to
    the light, under the reference's own visibility rule (1e-6 < t < dist - 1e-6)."""
    s, ref = load(root, "config/cbox.conf", width=8, height=8, spp=1)
    n_recs, n_pairs, rec_tri, blockers = s.plane_records()
    assert (blockers[:n_recs] != 0xFFFFFFFF).all() and (blockers[n_recs:] == 0xFFFFFFFF).all()
    # the room is convex: only the cube (records 8..13) and the luminaire cover can ever be in the way
    assert int(np.bitwise_or.reduce(blockers[:n_recs])) & 0x1F == 0
    verts = ref.verts.reshape(-1, 3, 3).astype(np.float64)
    tri_rec = np.full(26, -1)
    for bit, t in enumerate(rec_tri):
        if t >= 0:
            tri_rec[t] = bit // 2
    L = np.asarray(ref.lights[0][4:16]).reshape(4, 3)            # type, radiance[3], then the four corners
    rng = np.random.default_rng(5)
    n_blocked = 0
    for tri in range(26):
        for _ in range(60):
            w = rng.dirichlet([1, 1, 1])
            p2 = w @ verts[tri]                                 # point on the record
            x1, x2 = rng.random(2)
            p1 = L[0] + (L[1] - L[0]) * x1 + (L[2] - L[1]) * x2  # point on the light
            dist = np.linalg.norm(p1 - p2)
            d = (p2 - p1) / dist
            for other in range(26):
                t = oracle.tri_raycast_t(p1, d, *verts[other])
                if np.isfinite(t) and 1e-6 < t < dist - 1e-6:
                    n_blocked += 1
                    assert blockers[tri_rec[tri]] >> tri_rec[other] & 1, (tri, other)
    assert n_blocked > 100                                       # the test did see blocked segments


def test_scenes_without_a_single_square_light_keep_every_blocker(root):
    s, ref = load(root, "config/cbox_cube.conf", width=8, height=8, spp=1)   # two lights
    n_recs, n_pairs, rec_tri, blockers = s.plane_records()
    assert n_recs <= 32 and (blockers == 0xFFFFFFFF).all()
