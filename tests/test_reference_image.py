"""The one picture the reference holds of a scene whose assets are in its checkout --
results/cbox_cube.png, a window screenshot of config/cbox_cube.conf (README.md:36) -- against the
oracle (CPU) and the GPU library rendering this repository's copy of that conf
(config/cbox_cube_ref.conf: the reference's keys, only ModelName and ResultPath differ; the mesh is
the reference's model/cbox/cbox_test_1.obj array for array, tests/test_scene_io.py).

Two kinds of statement, fixture tests/golden/cbox_cube_ref64.npz (tools/extract_cbox_cube_fixture.py):

* EXACT, reference-sourced: which pixels the reference's camera rays hit anything at, and which hit
  the emitter (saturated white), at 512 x 512 -- 244 357 and 1 470 pixels.  Both sets must equal the
  first-hit map to the pixel.  This pins camera matrix, triangulation and intersector at pixel level
  (not at ulp level).
* LOOSE: a screenshot, of unknown sample count and conf revision, colour-managed by the viewer.  The
  picture is compared after undoing the embedded display profile, pooled 8 x 8 to 64 x 64, as
  per-channel means and a low-frequency RMS difference, inside and outside the glass cube.  The one
  tracer parameter that was scanned is Depth (3, 5, 7, 10, 15; nothing in the restatement was
  touched): with the checkout's Depth 5 the cube is 7 % darker than the screenshot's (paths through
  glass run out of bounces), from Depth 10 on the means agree to 0.1 % and the RMS difference is the
  noise of the two renders (measured: 0.0065 in the cube, 0.0051 elsewhere at 32 spp) -- the author's
  picture was taken at a larger Depth than the conf he committed.  The bars below are stated with
  the values measured when the test was written.  The scene has one sphere light, so what this
  checks is the unshadowed closed-form sphere term (cLight.cpp:188-255), the per-bounce clamp
  (cPathTracer.cpp:685-688), the pole-peaked direction sampler (cMathUtil.cpp:230-267), the Lambert
  term and the glass BSDF with its constant 1/1.5 (cBxDF.cpp:156-227); the square light's extra
  1/(2 pi) is not in this picture."""
import os

import numpy as np
import pytest

from oracle import oracle, scene_io

CONF = "config/cbox_cube_ref.conf"
CUBE = (slice(36, 62), slice(15, 39))       # the glass cube in the 64 x 64 pooled frame


def fixture(golden):
    g = np.load(os.path.join(golden, "cbox_cube_ref64.npz"))
    valid = np.zeros((512, 512), bool)
    r0, c0, r1, c1 = g["valid"]
    valid[r0:r1, c0:c1] = True
    hit = np.unpackbits(g["hit_bits"]).reshape(512, 512).astype(bool)
    lit = np.unpackbits(g["lit_bits"]).reshape(512, 512).astype(bool)
    return g, valid, hit, lit


def pooled(img_sum, finalize):
    """OutputImage's quantisation (int(c * 255), cPathTracer.cpp:265-272) of the clamped frame, then the
    fixture's 8 x 8 pooling."""
    q = np.floor(finalize(img_sum) * 255.0) / 255.0
    return q.reshape(64, 8, 64, 8, 3).mean(axis=(1, 3))


def compare(Q, g):
    S = g["rgb64"].astype(np.float64)
    inside = g["weight64"] > 0.99
    cube = np.zeros((64, 64), bool)
    cube[CUBE] = True
    out = {}
    for name, m in (("cube", cube & inside), ("rest", ~cube & inside)):
        d = (Q - S)[m]
        out[name] = (Q[m].mean(axis=0) / S[m].mean(axis=0), float(np.sqrt((d ** 2).mean())))
    return out


def check_silhouettes(ids, golden):
    g, valid, hit, lit = fixture(golden)
    ids = ids.reshape(512, 512)
    assert int(((ids >= 0) != hit)[valid].sum()) == 0           # 244 357 pixels hit something, the same ones
    assert int((((ids == 10) | (ids == 11)) != lit)[valid].sum()) == 0   # 1 470 see the emitter, the same ones
    assert int(hit.sum()) == 244357 and int(lit.sum()) == 1470


def check_picture(render, golden):
    """render(depth, spp) -> pooled 64 x 64 frame.  Bars: (largest |mean ratio - 1| over the channels, RMS)."""
    g = fixture(golden)[0]
    # measured when written (oracle, 16 spp): depth 10 -> cube ratios within 0.004, RMS 0.0080; rest within 0.002, RMS 0.0064
    r = compare(render(10, 16), g)
    assert np.abs(r["cube"][0] - 1).max() < 0.02 and r["cube"][1] < 0.015, r
    assert np.abs(r["rest"][0] - 1).max() < 0.02 and r["rest"][1] < 0.012, r
    # the conf as committed (Depth 5): measured cube ratios (0.929, 1.019, 0.968), RMS 0.040; rest (0.974, 1.014, 0.986), RMS 0.013
    r5 = compare(render(5, 16), g)
    assert np.abs(r5["cube"][0] - 1).max() < 0.10 and r5["cube"][1] < 0.06, r5
    assert np.abs(r5["rest"][0] - 1).max() < 0.04 and r5["rest"][1] < 0.02, r5
    # and the scan has a direction: the deeper render is the closer one, in the cube above all
    assert r["cube"][1] < 0.5 * r5["cube"][1] and r["rest"][1] < r5["rest"][1]


def test_conf_is_the_references(root):
    """Every key the tracer consumes has the value the reference's config/cbox_cube.conf holds (checked
    against the reference when it is mounted; the values are also written out here)."""
    import json
    mine = json.load(open(os.path.join(root, CONF)))
    assert mine["Light"] == {"LightNum": 1, "LightType": ["sphere"], "LightInfo": [[300, 300, 300, 50]], "LightRadiance": [[20, 20, 20]]}
    assert (mine["PathTracer"]["Depth"], mine["PathTracer"]["Samples"], mine["MainWindowInfo"]["Width"]) == (5, 10, 512)
    theirs_path = "/root/reference/config/cbox_cube.conf"
    if os.path.exists(theirs_path):
        theirs = json.load(open(theirs_path))                   # duplicate "Light_bak" keys: the last wins, none is read
        for block in ("MainWindowInfo", "Camera", "Light", "PathTracer", "Scene"):
            for k, v in mine[block].items():
                if (block, k) in (("Scene", "ModelName"), ("PathTracer", "ResultPath")):
                    continue
                if isinstance(v, dict):
                    for k2, v2 in v.items():
                        assert theirs[block][k][k2] == v2, (block, k, k2)
                else:
                    assert theirs[block][k] == v, (block, k)


def test_reference_screenshot_silhouettes_equal_oracle_first_hits(root, golden):
    sc = scene_io.load_conf(os.path.join(root, CONF), root_dir=root)
    assert (sc.width, sc.height) == (512, 512)
    check_silhouettes(oracle.Oracle(sc).first_hit(), golden)


def test_oracle_reproduces_the_references_picture(root, golden):
    def render(depth, spp):
        sc = scene_io.load_conf(os.path.join(root, CONF), root_dir=root, depth=depth)
        return pooled(oracle.Oracle(sc, seed=13).render(spp=spp), oracle.finalize)
    check_picture(render, golden)


@pytest.mark.gpu
def test_gpu_reproduces_the_references_picture(root, golden):
    """The same two statements through the C ABI (ptx_first_hit, ptx_render in f64)."""
    import pathtracer_amd as pt
    s = pt.Scene.load_conf(os.path.join(root, CONF), root_dir=root)
    c = pt.Context(0)
    c.build_accel(s)
    check_silhouettes(c.first_hit(), golden)

    def render(depth, spp):
        img, _ = c.render(mode="f64", spp=spp, depth=depth, seed=13)
        return pooled(img, pt.finalize_image)
    check_picture(render, golden)
