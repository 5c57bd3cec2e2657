"""CPU tests of the C-ABI library: it loads, exports every symbol include/ptx.h declares, and
refuses to compute without a GPU (there is no CPU fallback in the product)."""
import ctypes
import os
import re

import numpy as np
import pytest

import pathtracer_amd as pt


def declared_symbols(root):
    text = open(os.path.join(root, "include/ptx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(root):
    names = declared_symbols(root)
    assert len(names) >= 24
    L = ctypes.CDLL(pt.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "missing export " + n
    # the Python binding covers the header one-to-one
    assert sorted(pt.SYMBOLS) == names


def test_header_cites_reference_for_every_entry_point(root):
    text = open(os.path.join(root, "include/ptx.h")).read()
    assert text.count("src/") >= 15
    assert "extern \"C\"" in text and "torch" not in text.lower()


def test_version_and_error_strings():
    L = pt.lib()
    assert b"gfx950" in L.ptx_version()
    assert L.ptx_scene_get_info(None, None) == 1
    assert b"null" in L.ptx_last_error()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(pt.PtxError) as e:
        pt.Context(0)
    assert e.value.code == 4 and "no CPU fallback" in e.value.message


def test_product_never_imports_the_oracle(root):
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pkg = os.path.join(root, "pathtracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_scene_from_arrays_roundtrip():
    verts = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float32)
    mats = np.zeros((1, 12)); mats[0, 3:6] = 0.5; mats[0, 9] = 1; mats[0, 10] = 1
    lights = np.zeros((1, 16)); lights[0, 0] = 1; lights[0, 1:4] = 1; lights[0, 4:8] = (0, 0, 3, 0.5)
    view = np.eye(4); view[2, 3] = -5
    s = pt.Scene.from_arrays(verts, None, [0], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    i = s.info
    assert (i.n_tris, i.n_mats, i.n_lights, i.width, i.height, i.depth, i.spp) == (1, 1, 1, 16, 8, 3, 2)
    assert np.array_equal(s.geometry()[0], verts)
    with pytest.raises(pt.PtxError):
        pt.Scene.from_arrays(verts, None, [3], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    with pytest.raises(pt.PtxError):        # below -1 (= no material)
        pt.Scene.from_arrays(verts, None, [-2], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    bad = verts.copy()
    bad[0, 1, 2] = np.inf
    with pytest.raises(pt.PtxError):
        pt.Scene.from_arrays(bad, None, [0], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    bad[0, 1, 2] = np.nan
    with pytest.raises(pt.PtxError):
        pt.Scene.from_arrays(bad, None, [0], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    s2 = pt.Scene.from_arrays(verts, None, [-1], mats, lights, view, (0, 0, 5), 45.0, 0.1, 16, 8, 3, 2)
    assert s2.geometry()[2].tolist() == [-1]


def test_obj_with_a_non_finite_vertex_is_a_parse_error(tmp_path):
    import json
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 inf 0\nf 1 2 3\n")
    conf = {"MainWindowInfo": {"Width": 8, "Height": 8},
            "Scene": {"ModelName": "m.obj", "WorldScale": 1.0, "ModelDisplacement": [0, 0, 0, 0]},
            "Camera": {"Type": "FPS", "CameraPos": [0, 0, 5], "CameraFront": [0, 0, -1], "CameraUp": [0, 1, 0],
                       "CameraFov": 45, "CameraNear": 0.1},
            "PathTracer": {"Depth": 2, "Samples": 1, "ResultPath": "o.ppm", "EnableIndirectLight": True,
                           "EnableBarycentricNormal": True, "DrawRegion": [0, 0, 8, 8], "RayCastAccel": {"Type": "None"}},
            "Light": {"LightNum": 0, "LightType": [], "LightInfo": [], "LightRadiance": []}}
    (tmp_path / "c.conf").write_text(json.dumps(conf))
    with pytest.raises(pt.PtxError) as e:
        pt.Scene.load_conf(str(tmp_path / "c.conf"), root_dir=str(tmp_path))
    assert "finite" in str(e.value)
