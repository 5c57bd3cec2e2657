"""GPU edge cases through the C ABI, against the oracle: table sizes beyond the LDS copies (many
materials, many lights), empty and degenerate geometry, no lights, extreme depths, and the hand-over
between the all-triangles intersector and the BVH at 64 / 65 triangles."""
import copy
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import oracle, scene_io
from test_gpu_parity import assert_f64_parity as _assert_f64_parity

pytestmark = pytest.mark.gpu
LAST = {}  # context and oracle of the latest render_both(): what a differing pixel is re-traced with
TRACER = {}  # options of the BVH tracer under test (set by the bvh_tracer fixture)


def assert_f64_parity(img, oimg, st=None, ost=None, depth=15):
    return _assert_f64_parity(img, oimg, st, ost, depth, ctx=LAST["ctx"], O=LAST["O"])


def gpu_scene(ref):
    return pt.Scene.from_arrays(ref.verts.astype(np.float32), ref.normals.astype(np.float32), ref.mat_ids, ref.mats, ref.lights,
                                ref.view, ref.cam_pos, ref.fov_deg, ref.near, ref.width, ref.height, ref.depth, ref.spp,
                                ref.enable_indirect, ref.enable_bary_normal)


def render_both(ref, **ctx_kw):
    c = pt.Context(0, **ctx_kw)
    for k, v in TRACER.items():
        c.set_option(k, v)
    c.build_accel(gpu_scene(ref))
    img, st = c.render(mode="f64", seed=13)
    O = oracle.Oracle(ref, seed=13)
    oimg, ost = O.render(stats=True)
    LAST.update(ctx=c, O=O)
    return img, st, oimg, ost, c


@pytest.fixture()
def cbox(root):
    return scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=40, height=40, spp=4)


def test_more_materials_than_the_lds_table_holds(cbox):
    ref = copy.deepcopy(cbox)
    n = ref.verts.shape[0]
    # one material per triangle, plus padding to 40 entries: same colours, so the image is unchanged
    mats = np.concatenate([ref.mats[ref.mat_ids], np.tile(ref.mats[:1], (40 - n, 1))])
    base, _, _, _, _ = render_both(cbox)
    ref.mats, ref.mat_ids = mats, np.arange(n, dtype=np.int32)
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost)
    assert np.array_equal(img, base)                      # tables from global memory: same bits as from LDS


def test_more_lights_than_the_lds_table_holds(cbox):
    ref = copy.deepcopy(cbox)
    rng = np.random.default_rng(3)
    lights = [ref.lights[0]]
    for k in range(34):                                   # 35 lights: squares and spheres mixed
        L = np.zeros(16)
        if k % 2:
            L[0] = 1.0                                    # sphere: centre + radius
            L[1:4] = rng.uniform(0.2, 1.0, 3)
            L[4:7] = rng.uniform(1.0, 4.5, 3)
            L[7] = rng.uniform(0.05, 0.3)
        else:
            L[0] = 0.0                                    # small square under the ceiling
            L[1:4] = rng.uniform(0.5, 3.0, 3)
            x, z = rng.uniform(0.5, 4.5, 2)
            L[4:16] = np.array([[x, 5.4, z], [x + 0.3, 5.4, z], [x + 0.3, 5.4, z + 0.3], [x, 5.4, z + 0.3]]).ravel()
        lights.append(L)
    ref.lights = np.array(lights)
    ref.depth = 4
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost, depth=4)
    assert st["shadow_rays"] > 0


def test_no_lights_and_no_triangles(cbox):
    ref = copy.deepcopy(cbox)
    ref.lights = np.zeros((0, 16))
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost)
    assert st["shadow_rays"] == 0
    empty = copy.deepcopy(cbox)
    empty.verts, empty.normals = np.zeros((0, 3, 3)), np.zeros((0, 3, 3))
    empty.mat_ids = np.zeros(0, dtype=np.int32)
    img, st, oimg, ost, _ = render_both(empty)
    assert (img == 0).all() and (oimg == 0).all()
    assert st["paths"] == 40 * 40 * 4 and st["segments"] == st["paths"] and st["records"] == 0


def test_degenerate_and_duplicate_triangles(cbox):
    ref = copy.deepcopy(cbox)
    v = ref.verts.copy()
    extra = np.stack([np.stack([v[6][0], v[6][0], v[6][1]]),          # zero area: two equal corners
                      np.stack([v[6][0], v[6][0], v[6][0]]),          # a point
                      v[7]])                                          # exact duplicate of face 7 (tie -> lowest id)
    ref.verts = np.concatenate([v, extra])
    ref.normals = np.concatenate([ref.normals, ref.normals[[6, 6, 7]]])
    ref.mat_ids = np.concatenate([ref.mat_ids, ref.mat_ids[[6, 6, 7]]]).astype(np.int32)
    img, st, oimg, ost, c = render_both(ref)
    assert_f64_parity(img, oimg, st, ost)
    assert np.array_equal(c.first_hit(), oracle.Oracle(ref).first_hit())


@pytest.mark.parametrize("depth", [1, 2, 40])
def test_extreme_depths(cbox, depth):
    ref = copy.deepcopy(cbox)
    ref.depth = depth
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost, depth=depth)


def test_intersector_hand_over_at_the_record_limit(cbox):
    """The all-triangles intersector takes scenes of up to 64 triangles in up to 32 plane records:
    the Cornell box (14 records) plus 18 quads still uses it, plus 19 quads goes to the BVH; both
    agree with the oracle."""
    for quads in (18, 19):
        ref = copy.deepcopy(cbox)
        rng = np.random.default_rng(quads)
        c0 = rng.uniform(1.0, 4.5, (quads, 3))
        e1 = rng.uniform(-0.3, 0.3, (quads, 3))
        e2 = np.cross(e1, rng.uniform(-1, 1, (quads, 3)))
        e2 *= 0.3 / np.linalg.norm(e2, axis=1, keepdims=True)
        # axis-parallel rectangles in float so that the four corners are exactly coplanar
        e1[:, 1:] = 0.0
        e2[:, 0] = 0.0
        f = lambda x: x.astype(np.float32).astype(np.float64)
        p0, p1, p3 = f(c0), f(f(c0) + f(e1)), f(f(c0) + f(e2))
        p2 = f(p1 + (p3 - p0))
        tris = np.stack([np.stack([p0, p1, p2], 1), np.stack([p0, p2, p3], 1)], 1).reshape(-1, 3, 3)
        nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        ref.verts = np.concatenate([ref.verts, tris])
        ref.normals = np.concatenate([ref.normals, f(np.repeat(nrm[:, None, :], 3, axis=1))])
        ref.mat_ids = np.concatenate([ref.mat_ids, np.repeat(rng.integers(0, len(ref.mats), quads), 2)]).astype(np.int32)
        img, st, oimg, ost, c = render_both(ref)
        n_recs, n_pairs, _, _ = gpu_scene(ref).plane_records()
        assert c.accel_info()["brute_force"] == (n_recs <= 32)
        assert (quads == 18) == (n_recs <= 32)
        assert_f64_parity(img, oimg, st, ost)
        assert np.array_equal(c.first_hit(), oracle.Oracle(ref).first_hit())


# ---- BVH scenes without square lights: the ring tracer (hits and rays queue per wave) and the
# wavefront tracer (traversal and shading in separate kernels) ----------------------------------------
@pytest.fixture()
def mis(root):
    return scene_io.load_conf(os.path.join(root, "config/mis.conf"), root_dir=root, width=48, height=32, spp=3)


@pytest.fixture(params=["rings", "wavefront"])
def bvh_tracer(request):
    TRACER.clear()
    TRACER.update(bvh_wavefront=int(request.param == "wavefront"))
    yield request.param
    TRACER.clear()


def assert_tracer(c, bvh_tracer):
    assert ("k_wf_" in c.last_kernel()) == (bvh_tracer == "wavefront"), c.last_kernel()


@pytest.mark.parametrize("depth", [1, 2, 40])
def test_ring_tracer_extreme_depths(mis, depth, bvh_tracer):
    ref = copy.deepcopy(mis)
    ref.depth = depth
    img, st, oimg, ost, c = render_both(ref)
    assert not c.accel_info()["brute_force"]
    assert_tracer(c, bvh_tracer)
    assert_f64_parity(img, oimg, st, ost, depth=depth)


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (5, 3, 1), (9, 7, 33)])
def test_ring_tracer_fewer_paths_than_lanes_and_ragged_counts(root, w, h, spp, bvh_tracer):
    ref = scene_io.load_conf(os.path.join(root, "config/mis.conf"), root_dir=root, width=w, height=h, spp=spp)
    img, st, oimg, ost, c = render_both(ref)
    assert_tracer(c, bvh_tracer)
    assert st["paths"] == w * h * spp
    assert_f64_parity(img, oimg, st, ost)


def test_ring_tracer_without_lights_and_without_indirect_light(mis, bvh_tracer):
    ref = copy.deepcopy(mis)
    ref.lights = np.zeros((0, 16))
    img, st, oimg, ost, c = render_both(ref)
    assert_tracer(c, bvh_tracer)
    assert_f64_parity(img, oimg, st, ost)
    ref = copy.deepcopy(mis)
    ref.enable_indirect = False
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost)
    assert st["segments"] == st["paths"]                      # one segment per path: no continuation rays


def test_ring_tracer_with_tables_too_large_for_lds(mis, bvh_tracer):
    """More materials and more (sphere) lights than the LDS copies hold: the ring tracer reads both
    tables from global memory (the wavefront tracer's shading kernel always does); the material
    split alone must not change a bit."""
    base, _, _, _, _ = render_both(mis)
    ref = copy.deepcopy(mis)
    n_m = len(ref.mats)
    reps = 40 // n_m + 1
    ref.mats = np.tile(ref.mats, (reps, 1))                     # copies of the table ...
    ref.mat_ids = (ref.mat_ids + n_m * (np.arange(len(ref.mat_ids)) % reps)).astype(np.int32)  # ... used in turn
    img, st, oimg, ost, _ = render_both(ref)
    assert len(ref.mats) > 32
    assert_f64_parity(img, oimg, st, ost)
    assert np.array_equal(img, base)
    rng = np.random.default_rng(5)
    extra = []
    for k in range(30):
        L = np.zeros(ref.lights.shape[1])
        L[0] = 1.0
        L[1:4] = rng.uniform(0.05, 0.3, 3)
        L[4:7] = ref.lights[k % len(ref.lights)][4:7] + rng.uniform(-1, 1, 3)
        L[7] = rng.uniform(0.05, 0.2)
        extra.append(L)
    ref.lights = np.concatenate([ref.lights, np.array(extra)])
    assert len(ref.lights) > 32
    img, st, oimg, ost, _ = render_both(ref)
    assert_f64_parity(img, oimg, st, ost)


@pytest.mark.parametrize("position", [0, 3])
def test_ring_tracer_with_one_square_light_among_sphere_lights(mis, position, bvh_tracer):
    """The square light's visibility tests go round the rings (wavefront tracer: through the test queue,
    the sphere lights' terms in slots that count without a test); the light sum keeps the light order
    whether the square light comes first or in the middle."""
    ref = copy.deepcopy(mis)
    lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
    c, e, y = 0.5 * (lo + hi), 0.2 * (hi - lo), hi[1] - 0.05 * (hi[1] - lo[1])
    L = np.zeros(ref.lights.shape[1])
    L[1:4] = 5.0
    L[4:16] = np.array([[c[0] - e[0], y, c[2] - e[2]], [c[0] + e[0], y, c[2] - e[2]],
                        [c[0] + e[0], y, c[2] + e[2]], [c[0] - e[0], y, c[2] + e[2]]]).ravel()
    ref.lights = np.insert(ref.lights, position, L, axis=0)
    img, st, oimg, ost, c_ = render_both(ref)
    assert not c_.accel_info()["brute_force"]
    assert_tracer(c_, bvh_tracer)
    assert st["shadow_rays"] > 0
    assert_f64_parity(img, oimg, st, ost)


def test_wavefront_tracer_with_more_than_twelve_lights(mis):
    """Twenty lights, six of them square.  The physically based estimator tests every light: its
    visibility tests go through the wavefront tracer's test queue (a slot per light and segment, the
    tests named in dense lists per light; up to 32 lights: a segment marks its tests in a 32-bit set) and
    meet the oracle running that estimator.  The reference's estimator tests the squares only and stays
    on the ring tracer beyond twelve lights (every light would take a slot): equal to the lane tracer bit
    for bit, as ever."""
    ref = copy.deepcopy(mis)
    lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
    rng = np.random.default_rng(11)
    lights = list(ref.lights)
    for k in range(15):
        L = np.zeros(ref.lights.shape[1])
        L[1:4] = rng.uniform(0.5, 4.0, 3)
        if k % 5 < 2:                                                # a small square under the top of the scene
            x, z = rng.uniform(lo[[0, 2]] + 0.1 * (hi - lo)[[0, 2]], hi[[0, 2]] - 0.2 * (hi - lo)[[0, 2]])
            e, y = 0.08 * (hi - lo)[[0, 2]], hi[1] - rng.uniform(0.02, 0.1) * (hi[1] - lo[1])
            L[4:16] = np.array([[x, y, z], [x + e[0], y, z], [x + e[0], y, z + e[1]], [x, y, z + e[1]]]).ravel()
        else:
            L[0] = 1.0
            L[4:7] = rng.uniform(lo, hi)
            L[7] = rng.uniform(0.02, 0.1) * float((hi - lo).max())
        lights.insert(int(rng.integers(0, len(lights) + 1)), L)
    ref.lights = np.array(lights)
    assert len(ref.lights) == 20 and int((ref.lights[:, 0] == 0).sum()) == 6
    TRACER.clear()
    img, st, oimg, ost, c = render_both(ref)
    assert "k_trace_rings" in c.last_kernel(), c.last_kernel()
    assert st["shadow_rays"] > 0
    assert_f64_parity(img, oimg, st, ost)
    lane = pt.Context(0)
    lane.set_option("bvh_rings", 0)
    lane.build_accel(gpu_scene(ref))
    limg, lst = lane.render(mode="f64", seed=13)
    assert np.array_equal(limg, img)
    assert (lst["segments"], lst["shadow_rays"], lst["records"]) == (st["segments"], st["shadow_rays"], st["records"])
    # the physically based estimator: every light is tested, through the wavefront tracer's test queue
    F = oracle.Oracle(ref, seed=13, estimator="fast")
    fimg, fst = c.render(mode="f64", seed=13, estimator="fast")
    assert "k_wf_shade_fast" in c.last_kernel(), c.last_kernel()
    foimg, fost = F.render(stats=True)
    scale = np.maximum(1.0, np.abs(foimg).max(axis=-1, keepdims=True))
    _assert_f64_parity(fimg / scale, foimg / scale, fst, fost, depth=ref.depth, ctx=c, O=F, render_kw=dict(estimator="fast"))
    assert fst["shadow_rays"] > st["shadow_rays"]
    lfimg, lfst = lane.render(mode="f64", seed=13, estimator="fast")
    assert np.abs(lfimg - fimg).max() <= 1e-9 * max(1.0, float(np.abs(fimg).max()))
    assert (lfst["segments"], lfst["shadow_rays"], lfst["records"]) == (fst["segments"], fst["shadow_rays"], fst["records"])
    # thirty-three lights are one too many for the set: such a scene stays on the lane tracer
    ref.lights = np.array(list(ref.lights) + [ref.lights[-1]] * 13)
    c33 = pt.Context(0)
    c33.build_accel(gpu_scene(ref))
    c33.render(mode="f64", seed=13, spp=1, estimator="fast")
    assert "k_wf_" not in c33.last_kernel(), c33.last_kernel()


def test_ring_tracer_with_several_square_lights(mis, bvh_tracer):
    """Every square light with a valid sample costs the hit one more trip round the rings; results
    and test counts are those of the lane tracer and of the oracle."""
    ref = copy.deepcopy(mis)
    lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
    rng = np.random.default_rng(11)
    squares = []
    for k in range(4):
        c = lo + (hi - lo) * rng.uniform(0.2, 0.8, 3)
        e = 0.08 * (hi - lo)
        y = hi[1] - (0.05 + 0.1 * k) * (hi[1] - lo[1])
        L = np.zeros(ref.lights.shape[1])
        L[1:4] = rng.uniform(1.0, 6.0, 3)
        L[4:16] = np.array([[c[0] - e[0], y, c[2] - e[2]], [c[0] + e[0], y, c[2] - e[2]],
                            [c[0] + e[0], y, c[2] + e[2]], [c[0] - e[0], y, c[2] + e[2]]]).ravel()
        squares.append(L)
    lights = list(ref.lights)
    for k, L in enumerate(squares):
        lights.insert(2 * k, L)                                   # squares and spheres interleaved
    ref.lights = np.array(lights)
    img, st, oimg, ost, c_ = render_both(ref)
    assert not c_.accel_info()["brute_force"]
    assert_tracer(c_, bvh_tracer)                                  # nine lights, four of them square: within the wavefront tracer's twelve
    assert st["shadow_rays"] > st["segments"]                      # more than one test per segment on average
    assert_f64_parity(img, oimg, st, ost)
    lane = pt.Context(0)
    lane.set_option("bvh_rings", 0)
    lane.build_accel(gpu_scene(ref))
    limg, lst = lane.render(mode="f64", seed=13)
    assert np.array_equal(limg, img)
    assert (lst["segments"], lst["shadow_rays"], lst["records"]) == (st["segments"], st["shadow_rays"], st["records"])


def test_wavefront_tracer_at_great_depths(mis):
    """A depth far beyond the length of any path: the wavefront tracer stops launching once its ray
    queue is empty (same bits and counts as the ring tracer, which has no such loop); at a depth whose
    records do not fit the memory for even its smallest batch the scene goes to the ring tracer."""
    ref = copy.deepcopy(mis)
    scene = gpu_scene(ref)
    out = {}
    for wavefront in (0, 1):
        c = pt.Context(0)
        c.set_option("bvh_wavefront", wavefront)
        c.build_accel(scene)
        img, st = c.render(mode="f64", seed=13, depth=300)
        assert ("k_wf_" in c.last_kernel()) == bool(wavefront)
        out[wavefront] = (img, st)
        if wavefront:
            deep, std = c.render(mode="f64", seed=13, depth=65535)      # 48 B x 65535 per path: no batch of it fits
            assert "k_trace_rings" in c.last_kernel()
            assert std["segments"] >= st["segments"] and np.isfinite(deep).all()
            again, st2 = c.render(mode="f64", seed=13, depth=300)       # and back
            assert "k_wf_" in c.last_kernel() and np.array_equal(again, img) and st2["segments"] == st["segments"]
    assert np.array_equal(out[0][0], out[1][0])
    assert all(out[0][1][k] == out[1][1][k] for k in ("paths", "segments", "records"))
    assert out[0][1]["segments"] < 40 * out[0][1]["paths"]              # nowhere near 300 segments per path


def test_state_that_does_not_fit_shrinks_or_fails_cleanly(cbox):
    """The path state is sized against the memory there is: a depth whose record slots would take
    terabytes on the full grid runs on a smaller one, a request that cannot be met at all comes back
    as an error code, and either way the context renders the next frame as if nothing had happened."""
    ref = copy.deepcopy(cbox)
    c = pt.Context(0)
    c.build_accel(gpu_scene(ref))
    base, st0 = c.render(mode="f64", seed=13)
    deep, st = c.render(mode="f64", seed=13, depth=65535)            # 192 slots x 65535 bounces x 64 B per wave
    assert np.isfinite(deep).all() and st["paths"] == st0["paths"] and st["state_bytes"] > st0["state_bytes"]
    assert st["segments"] >= st0["segments"]
    again, st1 = c.render(mode="f64", seed=13)
    assert np.array_equal(again, base) and st1["segments"] == st0["segments"]
    with pytest.raises(pt.PtxError) as e:
        c.render(mode="f64", depth=70000)
    assert e.value.code == 1
    huge, _ = c.render(mode="f64", seed=13, batch_paths=1 << 40)   # a batch is never larger than the frame
    assert np.array_equal(huge, base)
    again, _ = c.render(mode="f64", seed=13)
    assert np.array_equal(again, base)
