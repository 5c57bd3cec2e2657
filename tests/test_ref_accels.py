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


def test_octree_cache_round_trip(cbox_rays, tmp_path):
    """cOctree::WriteOctree / ReadOctree (src/geometry/cAccelStruct.cpp:485-611): the text cache the
    reference keeps beside the mesh.  One line per node, breadth first: id, leaf flag, three
    (low, high) pairs at six significant digits, then eight child ids or the face ids, every item
    followed by a space.  Read back, the octree has the same nodes and faces and answers this
    scene's rays like the one that was built (its boxes differ in the seventh digit).  The reference
    ships no .octree file: the format is pinned by its writer's source alone."""
    ref, O, (o, d, ids, pts), (o2, d2, ids2, pts2) = cbox_rays
    A = oracle.RefAccel(ref.verts, "OCTREE", 8)
    path = tmp_path / "cbox.obj.octree"
    A.write_octree(str(path))
    lines = path.read_text().split("\n")
    assert lines[-1] == "" and len(lines) - 1 == A.info()["boxes"] == 49
    first = lines[0].split()
    assert first[:2] == ["0", "0"] and [float(x) for x in first[2:8]] == [-1e-05, 5.56001, -1e-05, 5.48801, -1e-05, 5.59201]
    assert first[8:] == [str(k) for k in range(1, 9)]                   # the root's children are the next eight nodes made
    assert all(ln.endswith(" ") for ln in lines[:-1])
    seen, faces = set(), 0
    for ln in lines[:-1]:
        f = ln.split()
        assert int(f[0]) not in seen
        seen.add(int(f[0]))
        assert all(float("%g" % float(x)) == float(x) for x in f[2:8])  # six significant digits
        if f[1] == "0":
            assert len(f) == 16
        else:
            faces += len(f) - 8
            assert all(0 <= int(x) < 26 for x in f[8:])
    assert faces == A.info()["face_refs"] == 157
    B = oracle.RefAccel.from_octree_file(ref.verts, str(path))
    assert B.info() == A.info()
    for ro, rd in ((o, d), (o2, d2)):
        a_ids, a_pts = A.raycast(ro, rd)
        b_ids, b_pts = B.raycast(ro, rd)
        assert np.array_equal(a_ids, b_ids) and np.array_equal(a_pts, b_pts, equal_nan=True)
    # written again from what was read, the file only differs where a bound had more digits to lose
    path2 = tmp_path / "again.octree"
    B.write_octree(str(path2))
    assert path2.read_text() == path.read_text()


def test_octree_cache_reader_errors(cbox_rays, tmp_path):
    ref = cbox_rays[0]
    with pytest.raises(RuntimeError, match="cannot open"):
        oracle.RefAccel.from_octree_file(ref.verts, str(tmp_path / "missing.octree"))
    p = tmp_path / "big.octree"
    p.write_text("0 0 0 1 0 1 0 1 1 2 3 4 5 6 7 1000001 \n")             # the reference prints "buffer exceed" and exits
    with pytest.raises(RuntimeError, match="buffer"):
        oracle.RefAccel.from_octree_file(ref.verts, str(p))
    p.write_text("0 1 0 1 0 1 0 1 3 99 \n")                               # a face the mesh does not have
    with pytest.raises(RuntimeError, match="malformed"):
        oracle.RefAccel.from_octree_file(ref.verts, str(p))
    p.write_text("1 1 0 1 0 1 0 1 3 \n")                                  # no node 0: nothing to start a query at
    with pytest.raises(RuntimeError, match="malformed"):
        oracle.RefAccel.from_octree_file(ref.verts, str(p))
    p.write_text("0 1  -1 6 -1 6 -1 6 " + " ".join(str(k) for k in range(26)) + " \n\n")
    one = oracle.RefAccel.from_octree_file(ref.verts, str(p))            # a single leaf with every face: the plain loop
    assert one.info()["boxes"] == 1 and one.info()["face_refs"] == 26
    with pytest.raises(RuntimeError):
        oracle.RefAccel(ref.verts, "AABB", 2).write_octree(str(tmp_path / "grid.octree"))
