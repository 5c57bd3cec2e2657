"""The reference's two culling structures (uniform grid `cAABB`, `cOctree`;
src/geometry/cAccelStruct.cpp:89-263, 266-482, 614-730 and the box tests of
src/geometry/cBaseMesh.cpp:60-223 of the reference) restated in oracle/ref_accels.c, held against the
plain loop over all faces that the oracle and the GPU core take as ground truth.  They run the same
exact ray/triangle test on a subset of the faces, so they can lose a hit or keep a farther one,
never find a closer one."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle, scene_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_accel_report  # noqa: E402


@pytest.fixture(scope="module")
def cbox_rays(root):
    ref = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=64, height=64, spp=1)
    O, cam, sec = ref_accel_report.rays_of(ref, 3000)
    o2, d2 = sec
    ids2 = np.empty(o2.shape[0], dtype=np.int32)
    pts2 = np.empty_like(o2)
    for i in range(o2.shape[0]):
        ids2[i], pts2[i] = O.closest_hit(o2[i], d2[i])
    return ref, O, cam, (o2, d2, ids2, pts2)


@pytest.mark.parametrize("kind,param,boxes,refs", [("AABB", 2, 8, 68), ("AABB", 4, 56, 182),
                                                  ("OCTREE", 30, 1, 26), ("OCTREE", 8, 49, 157)])
def test_structures_agree_with_the_plain_loop_on_the_cornell_box(cbox_rays, kind, param, boxes, refs):
    ref, O, (o, d, ids, pts), (o2, d2, ids2, pts2) = cbox_rays
    A = oracle.RefAccel(ref.verts, kind, param)
    info = A.info()
    # config/cbox.conf asks for OCTREE with capacity 30: 26 faces never fill the root, one leaf
    assert (info["boxes"], info["face_refs"]) == (boxes, refs)
    assert info["stranded_refs"] == 0 and info["unreachable_faces"] == 0 and info["reference_would_exit"] == 0
    for (ro, rd, want_ids, want_pts) in ((o, d, ids, pts), (o2, d2, ids2, pts2)):
        got_ids, got_pts = A.raycast(ro, rd)
        c = ref_accel_report.compare(want_ids, want_pts, got_ids, got_pts, ro)
        assert c["closer_than_plain_loop"] == 0 and c["same_face_other_point"] == 0
        assert c["different_face"] == 0          # measured: no ray of this scene is answered differently


def test_vistest_agrees_with_the_plain_loop(cbox_rays):
    ref, O, (o, d, ids, pts), _ = cbox_rays
    rng = np.random.default_rng(7)
    hit = np.flatnonzero(ids >= 0)
    a = pts[rng.choice(hit, 1500)]
    b = pts[rng.choice(hit, 1500)]
    b[:500] = [2.78, 5.48, 2.795]                # towards the light
    want = np.array([O.vistest(a[i], b[i]) for i in range(a.shape[0])])
    assert 0.05 < want.mean() < 0.95
    for kind, param in (("AABB", 2), ("AABB", 4), ("OCTREE", 30), ("OCTREE", 8)):
        got = oracle.RefAccel(ref.verts, kind, param).vistest(a, b)
        assert np.array_equal(got, want), (kind, param)


def test_octree_split_that_cannot_end_is_reported():
    # `capacity` faces round one vertex: every split hands all of them to the child at that vertex
    fan = []
    for k in range(6):                           # six sevenths of a disc: the vertex is on no split plane
        a0, a1 = 2 * np.pi * k / 7, 2 * np.pi * (k + 1) / 7
        fan.append([[0.3, 0.3, 0.3], [0.3 + np.cos(a0), 0.3 + np.sin(a0), 0.3 + 0.5 * np.sin(3 * a0)],
                    [0.3 + np.cos(a1), 0.3 + np.sin(a1), 0.3 + 0.5 * np.sin(3 * a1)]])
    verts = np.array(fan, dtype=np.float32).astype(np.float64)
    with pytest.raises(RuntimeError):
        oracle.RefAccel(verts, "OCTREE", 6)
    A = oracle.RefAccel(verts, "OCTREE", 7)      # one more than the fan holds: a single leaf
    assert A.info()["boxes"] == 1
    ids, pts = A.raycast([[0.5, 0.4, 5.0]], [[0.0, 0.0, -1.0]])
    assert ids[0] == 0 and abs(pts[0, 0] - 0.5) < 1e-12 and abs(pts[0, 1] - 0.4) < 1e-12


def test_axis_aligned_face_in_a_split_plane_meets_no_child():
    # the nine edge axes of the box/face test treat touching as apart: a flat face lying exactly in
    # the plane that halves its node belongs to neither half, where the reference prints
    # "[error] tOctreeNode::split add face ... failed" and exits (cAccelStruct.cpp:355-370)
    quads = []
    for k in range(4):
        x = float(k)
        quads += [[[x, 0, 0.5], [x + 1, 0, 0.5], [x + 1, 1, 0.5]], [[x, 0, 0.5], [x + 1, 1, 0.5], [x, 1, 0.5]]]
    verts = np.array(quads, dtype=np.float64)
    A = oracle.RefAccel(verts, "OCTREE", 8)
    i = A.info()
    assert i["reference_would_exit"] == 8 and i["unreachable_faces"] == 8
    assert oracle.RefAccel(verts, "OCTREE", 9).info()["reference_would_exit"] == 0


def test_bounds_start_from_the_smallest_positive_double():
    # a mesh wholly at negative coordinates: the reference's upper bound stays at ~0 (+1e-5), the
    # boxes reach up to the origin, and rays are still answered
    tri = np.array([[[-3, -3, -3], [-2, -3, -3], [-3, -2, -3]]], dtype=np.float64)
    A = oracle.RefAccel(tri, "AABB", 2)
    assert A.info()["boxes"] == 1 and A.info()["face_refs"] == 1     # only the far corner box holds it
    ids, pts = A.raycast([[-2.8, -2.8, 4.0]], [[0.0, 0.0, -1.0]])
    assert ids[0] == 0 and np.allclose(pts[0], [-2.8, -2.8, -3.0])


def test_report_runs(root):
    r = ref_accel_report.report("config/cbox.conf", 16, 100, structures=(("OCTREE", 30),))
    s = r["structures"]["OCTREE 30"]
    assert r["faces"] == 26 and s["camera"]["rays"] == 256 and s["camera"]["different_face"] == 0


def test_room_stand_in_loads_into_the_reference_octree(root, room_scene):
    # config/room.conf names OCTREE with capacity 90; the stand-in must be a mesh that builder can
    # take (no `capacity` faces round one point, no face in a split plane) and is answered alike
    ref = scene_io.load_conf(os.path.join(root, "config/room.conf"), root_dir=root, width=32, height=32, spp=1)
    ref.use_bvh = True
    v = ref.verts.reshape(-1, 3, 3)
    assert (np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 1]), axis=1) > 0).all()
    A = oracle.RefAccel(ref.verts, "OCTREE", 90)
    info = A.info()
    assert info["reference_would_exit"] == 0 and info["unreachable_faces"] == 0 and info["stranded_refs"] == 0
    O = oracle.Oracle(ref, seed=13)
    o, d = O.generate_rays()
    ids, pts = O.first_hit(with_points=True)
    got_ids, got_pts = A.raycast(o, d)
    assert np.array_equal(got_ids, ids) and np.array_equal(got_pts[ids >= 0], pts[ids >= 0])
