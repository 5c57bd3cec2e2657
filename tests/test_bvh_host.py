"""CPU tests of the host-built 4-wide BVH (ptx_scene_bvh_check): every triangle referenced exactly
once, child boxes nested, triangles strictly inside their boxes, and the stack bound reported by the
builder holds on every root-to-leaf path -- for the SAH tree and for the median-split fallback; with
reference splitting a triangle hangs in the tree through several references whose boxes cover it."""
import os

import pytest

import pathtracer_amd as pt


@pytest.mark.parametrize("median", [False, True])
@pytest.mark.parametrize("conf", ["config/cbox.conf", "config/mis.conf", "config/room.conf"])
def test_bvh_invariants(root, conf, median):
    s = pt.Scene.load_conf(os.path.join(root, conf), root_dir=root, width=8, height=8, spp=1)
    info = s.bvh_check(median=median)
    n = s.info.n_tris
    assert info["violations"] == 0
    assert info["triangle_refs"] == n
    # children = triangles + every node but the root
    assert info["n_children"] == n + info["n_nodes"] - 1
    assert info["n_nodes"] <= max(1, n)                      # far fewer nodes than a binary tree would have
    assert 0 < info["stack_need"] <= 3 * (info["max_depth"] + 1)
    assert info["stack_need"] <= 96                          # what the kernels can hold in LDS


def test_median_tree_is_shallower_than_sah_on_a_large_scene(root):
    s = pt.Scene.load_conf(os.path.join(root, "config/room.conf"), root_dir=root, width=8, height=8, spp=1)
    sah, med = s.bvh_check(False), s.bvh_check(True)
    assert med["max_depth"] <= sah["max_depth"]
    assert med["stack_need"] <= sah["stack_need"]


def test_tiny_and_degenerate_scenes(root):
    """Few triangles make a single node; identical triangles (all centroids equal: no split plane)
    are split by index."""
    import numpy as np
    from oracle import scene_io
    ref = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=4, height=4, spp=1)

    def scene_of(verts):
        n = verts.shape[0]
        return pt.Scene.from_arrays(verts.astype(np.float32), np.zeros_like(verts, dtype=np.float32), np.zeros(n, dtype=np.int32),
                                    ref.mats, ref.lights, ref.view, ref.cam_pos, ref.fov_deg, ref.near, 4, 4, 2, 1)

    tri = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.5]])
    for n in (1, 3, 4, 5, 9):
        verts = np.stack([tri + np.array([2.0 * i, 0.0, 0.0]) for i in range(n)])
        info = scene_of(verts).bvh_check()
        assert info["violations"] == 0 and info["triangle_refs"] == n
        if n <= 4:
            assert info["n_nodes"] == 1 and info["stack_need"] == n - 1
    verts = np.stack([tri] * 37)                               # 37 copies of one triangle
    for median in (False, True):
        info = scene_of(verts).bvh_check(median=median)
        assert info["violations"] == 0 and info["triangle_refs"] == 37


@pytest.mark.parametrize("conf", ["config/cbox.conf", "config/mis.conf", "config/room.conf"])
def test_reference_splitting_keeps_the_invariants(root, conf):
    """Reference splitting (off by default, option bvh_split): more references than triangles, every
    reference in the tree once, every triangle covered by the boxes of its references (a grid of points
    over it, each strictly inside one of them), child boxes nested, stack bound kept."""
    s = pt.Scene.load_conf(os.path.join(root, conf), root_dir=root, width=8, height=8, spp=1)
    plain, cut = s.bvh_check(), s.bvh_check(split=True)
    n = s.info.n_tris
    assert cut["violations"] == 0
    assert n < cut["triangle_refs"] <= n + int(0.3 * n) + 1 or n < 16
    assert cut["n_children"] == cut["triangle_refs"] + cut["n_nodes"] - 1
    assert cut["stack_need"] <= 96
    assert plain["triangle_refs"] == n
