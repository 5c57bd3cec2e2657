"""GPU parity tests (run on an MI355X with -m gpu).  Everything goes through the C ABI of
libptx_amd.so and is compared with the CPU oracle on the same seeded inputs.

Bars: camera rays, first-hit face ids, hit points and visibility flags are BIT-EXACT.  Images in
f64 mode use the reference's arithmetic with the same random numbers, so they agree with the oracle
up to the last-ulp differences between the device's and glibc's pow/acos/sin/cos: tolerance
1e-9 absolute on the pre-clamp sums with identical ray counts.  Images in f32 mode must satisfy the
relative-MSE bar  rMSE <= max(2 x rMSE(oracle seed A, oracle seed B), 1e-3)  at equal spp."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import oracle, parity, scene_io

pytestmark = pytest.mark.gpu
F64_TOL = parity.F64_TOL


OUTLIERS = []        # (test id, differing pixels, pixels, proven ties) of every comparison, printed at the end of the session


def assert_f64_parity(img, oimg, st=None, ost=None, depth=15, ctx=None, O=None, spp=None, origin=(0, 0),
                      sample_offset=0, render_kw=None):
    """f64 images must agree to F64_TOL in every pixel -- except where the path hinges on an exact
    geometric tie, and every exception is MATCHED to one (oracle/parity.py): for each differing
    pixel, each of its samples is rendered alone on the GPU (through the C ABI) and re-traced by the
    oracle; a sample whose two values differ must equal, to F64_TOL, the oracle's value with the
    closest-hit ties on its path (winner and runner-up within 1e-10) falling another, named way.
    That a tie exists on the path is not enough.  A sample without ties may instead be an amplified
    rounding: its difference within 1024 ulps times the factor by which the oracle measures the path to
    multiply an error of its directions (counted and reported apart).  Without ctx and O no exception
    is allowed at all.  origin: (row, column) of img[0, 0] in the frame ctx renders.  Returns the number of
    differing pixels."""
    img, oimg = np.asarray(img), np.asarray(oimg)
    bad, diff = parity.outliers(img, oimg)
    n_bad = len(bad)
    proven = amplified = 0
    if n_bad:
        assert ctx is not None and O is not None, "%d of %d pixels differ by more than %g" % (n_bad, diff.size, F64_TOL)
        n_spp = int(spp if spp is not None else O.c.spp)
        # a guard against proving thousands of pixels one by one (about 1 path in 2000 meets a tie in the Cornell box)
        assert n_bad <= max(2, diff.size * n_spp // 500), "%d of %d pixels differ: too many to be ties" % (n_bad, diff.size)
        r = parity.prove_pixels(ctx, O, [(int(y) + origin[0], int(x) + origin[1]) for y, x in bad],
                                range(sample_offset, sample_offset + n_spp), seed=int(O.c.seed), render_kw=render_kw)
        assert not r["unmatched"], "samples that differ from the oracle and equal no tie alternative of their path: %s" % (r["unmatched"][:3],)
        assert r["no_sample_differs"] == 0, "%d pixels differ although none of their samples does" % r["no_sample_differs"]
        proven = r["proven_ties"]
        assert proven + r["amplified"] == n_bad
        amplified = r["amplified"]
    OUTLIERS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], n_bad, int(diff.size), proven, amplified))
    assert oracle.rmse_rel(img, oimg) < 1e-4
    if st is not None:
        assert st["paths"] == ost["paths"]
        assert abs(int(st["segments"]) - int(ost["segments"])) <= depth * n_bad
    return n_bad


def load(root, conf, **kw):
    s = pt.Scene.load_conf(os.path.join(root, conf), root_dir=root, **kw)
    ref = scene_io.load_conf(os.path.join(root, conf), root_dir=root, **kw)
    return s, ref


def make_ctx(scene, **kw):
    c = pt.Context(0, **kw)
    c.build_accel(scene)
    return c


@pytest.fixture(scope="module")
def cbox64(root):
    s, ref = load(root, "config/cbox.conf", width=64, height=64, spp=8)
    return s, ref, make_ctx(s), oracle.Oracle(ref, seed=13)


def test_camera_rays_bit_exact(cbox64):
    s, ref, c, O = cbox64
    o, d = c.generate_rays()
    oo, od = O.generate_rays()
    assert np.array_equal(o, oo) and np.array_equal(d, od)


def test_first_hit_ids_bit_exact_small(cbox64, golden):
    s, ref, c, O = cbox64
    ids, pts = c.first_hit(with_points=True)
    oids, opts = O.first_hit(with_points=True)
    assert np.array_equal(ids, oids)
    assert np.array_equal(pts, opts, equal_nan=True)
    g = np.load(os.path.join(golden, "cbox_first_hit.npz"))
    assert np.array_equal(ids, g["ids_64"].astype(np.int32))


@pytest.mark.parametrize("w", [256, 1024])
def test_first_hit_ids_bit_exact_full_size(root, golden, w):
    s, ref = load(root, "config/cbox.conf", width=w, height=w, spp=1)
    c = make_ctx(s)
    ids, pts = c.first_hit(with_points=True)
    oids, opts = oracle.Oracle(ref).first_hit(with_points=True)
    assert int((ids != oids).sum()) == 0
    assert np.array_equal(pts, opts, equal_nan=True)
    if w == 256:
        g = np.load(os.path.join(golden, "cbox_first_hit.npz"))
        assert np.array_equal(ids, g["ids_256"].astype(np.int32))


def random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(0.05, 5.4, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a share of rays start ON surfaces (the secondary-ray situation)
    o[: n // 4, 1] = 0.0
    d[: n // 4, 1] = np.abs(d[: n // 4, 1])
    return o, d


def test_raycast_and_vistest_operators_bit_exact(cbox64):
    s, ref, c, O = cbox64
    o, d = random_rays(20000, 1)
    ids, pts = c.raycast(o, d)
    for i in range(0, 20000, 7):                      # oracle is scalar: check a strided subset
        f, p = O.closest_hit(o[i], d[i])
        assert f == ids[i] and np.array_equal(p, pts[i], equal_nan=True)
    rng = np.random.default_rng(2)
    p1 = np.tile(np.array([2.78, 5.48, 2.795]), (6000, 1)) + rng.uniform(-0.6, 0.6, (6000, 3)) * [1, 0, 1]
    p2 = rng.uniform(0.0, 5.4, (6000, 3))
    p2[:2000, 1] = 0.0
    p2[5990:] = p1[5990:]                            # zero-length segments are visible
    vis = c.vistest(p1, p2)
    ovis = np.array([O.vistest(a, b) for a, b in zip(p1, p2)])
    assert np.array_equal(vis, ovis) and 0.05 < vis.mean() < 0.98


@pytest.mark.parametrize("conf", ["config/cbox.conf", "config/mis.conf"])
def test_raycast_takes_directions_of_any_length(root, conf):
    """cAccelStruct::RayCast does not normalise the direction (tRay::Init keeps it as given): hits,
    points and ties are those of the unit ray whatever |d| is -- on the all-triangles intersector
    (Cornell box) and through the BVH (mis), whose culling bounds are distances."""
    s, ref = load(root, conf, width=32, height=32, spp=1)
    c, O = make_ctx(s), oracle.Oracle(ref, seed=13)
    lo, hi = ref.verts.reshape(-1, 3).min(0), ref.verts.reshape(-1, 3).max(0)
    rng = np.random.default_rng(7)
    n = 6000
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    unit_ids, unit_pts = c.raycast(o, d)
    assert (unit_ids >= 0).mean() > 0.3
    for scale in (1e-3, 0.1, 10.0, 1e4):
        ids, pts = c.raycast(o, d * scale)
        for i in range(0, n, 5):
            f, p = O.closest_hit(o[i], d[i] * scale)
            assert f == ids[i] and np.array_equal(p, pts[i], equal_nan=True), (scale, i)
        # the same faces as at unit length, except where two candidates are a few ulps apart
        assert (ids != unit_ids).mean() < 1e-3
    ids, _ = c.raycast(o[:64], np.zeros((64, 3)))                       # no direction: no hit (NaN everywhere)
    assert (ids == -1).all()


def test_bvh_traversal_equals_brute_force(root):
    s, ref = load(root, "config/cbox.conf", width=128, height=128, spp=4)
    brute, bvh = make_ctx(s), make_ctx(s, brute_force_limit=0)
    assert brute.accel_info()["brute_force"] and not bvh.accel_info()["brute_force"]
    a, pa = brute.first_hit(with_points=True)
    b, pb = bvh.first_hit(with_points=True)
    assert np.array_equal(a, b) and np.array_equal(pa, pb, equal_nan=True)
    o, d = random_rays(50000, 3)
    ia, qa = brute.raycast(o, d)
    ib, qb = bvh.raycast(o, d)
    assert np.array_equal(ia, ib) and np.array_equal(qa, qb, equal_nan=True)
    ia_, sa = brute.render(mode="f64")
    ib_, sb = bvh.render(mode="f64")
    assert np.array_equal(ia_, ib_) and sa["segments"] == sb["segments"]


def test_render_f64_matches_golden_image(root, golden):
    g = np.load(os.path.join(golden, "cbox_32_spp16.npz"))
    s, ref = load(root, "config/cbox.conf", width=32, height=32, spp=16)
    c = make_ctx(s)
    img, st = c.render(mode="f64", seed=13)
    ost = dict(paths=int(g["stats"][0]), segments=int(g["stats"][1]))
    assert_f64_parity(img, g["img"], st, ost, ctx=c, O=oracle.Oracle(ref, seed=13))


@pytest.mark.parametrize("conf,w,h,spp", [("config/cbox.conf", 64, 64, 8), ("config/cbox_glossy.conf", 48, 48, 8),
                                          ("config/cbox_cube.conf", 48, 48, 8), ("config/mis.conf", 96, 64, 4),
                                          ("config/room.conf", 96, 64, 2)])
def test_render_f64_matches_oracle(root, conf, w, h, spp):
    s, ref = load(root, conf, width=w, height=h, spp=spp)
    c, O = make_ctx(s), oracle.Oracle(ref, seed=13)
    img, st = c.render(mode="f64", seed=13)
    oimg, ost = O.render(stats=True)
    assert st["paths"] == w * h * spp
    assert_f64_parity(img, oimg, st, ost, depth=ref.depth, ctx=c, O=O)


def test_render_flags_and_region(root):
    for kw in (dict(enable_indirect=False), dict(enable_bary_normal=False), dict(region=(5, 7, 20, 30))):
        s, ref = load(root, "config/cbox.conf", width=40, height=40, spp=4)
        for k, v in kw.items():
            setattr(ref, k, v)
        verts, normals, mat_ids = s.geometry()
        mats, lights = s.tables()
        s2 = pt.Scene.from_arrays(verts, normals, mat_ids, mats, lights, ref.view, ref.cam_pos, ref.fov_deg, ref.near,
                                  40, 40, ref.depth, 4, ref.enable_indirect, ref.enable_bary_normal)
        c = make_ctx(s2)
        rows = None
        O = oracle.Oracle(ref, seed=13)
        oimg = O.render()
        if "region" in kw:
            # rows through the render options; columns are checked by masking the oracle image
            rows = (5, 21)
            ref.region = (5, 0, 20, 40)
            O = oracle.Oracle(ref, seed=13)
            oimg = O.render()
        img, st = c.render(mode="f64", rows=rows)
        assert_f64_parity(img, oimg, ctx=c, O=O)
        if "region" in kw:
            assert (img[:5] == 0).all() and (img[21:] == 0).all()


@pytest.mark.parametrize("name,estimator", [("cbox", "reference"), ("mis", "reference"), ("mis", "fast")])
def test_draw_region_from_conf(root, tmp_path, name, estimator):
    """DrawRegion sub-window (rows 3..17, columns 5..29, inclusive): the all-triangles tracer on the
    Cornell box, the wavefront tracer on mis under both estimators; pixels outside stay 0."""
    import json
    conf = json.load(open(os.path.join(root, "config/%s.conf" % name)))
    conf["MainWindowInfo"]["Width"] = conf["MainWindowInfo"]["Height"] = 36
    conf["PathTracer"]["DrawRegion"] = [3, 5, 17, 29]
    conf["PathTracer"]["Samples"] = 3
    p = tmp_path / "r.conf"
    p.write_text(json.dumps(conf))
    s = pt.Scene.load_conf(str(p), root_dir=root)
    ref = scene_io.load_conf(str(p), root_dir=root, full_region=False)
    assert tuple(ref.region) == (3, 5, 17, 29)
    c, O = make_ctx(s), oracle.Oracle(ref, seed=13, estimator=estimator)
    img, st = c.render(mode="f64", estimator=estimator)
    assert ("k_wf_" in c.last_kernel()) == (name == "mis")
    assert st["paths"] == 15 * 25 * 3
    oimg = O.render()
    scale = np.maximum(1.0, np.abs(oimg).max(axis=-1, keepdims=True))
    assert_f64_parity(img / scale, oimg / scale, ctx=c, O=O, render_kw=dict(estimator=estimator))
    mask = np.zeros((36, 36), bool)
    mask[3:18, 5:30] = True
    assert (img[~mask] == 0).all()
    # rows of the window in two pieces and a batch per few pixels add up to the same sums
    top, _ = c.render(mode="f64", estimator=estimator, rows=(0, 10), batch_paths=50)
    both, _ = c.render(mode="f64", estimator=estimator, rows=(10, 36), accumulate=True, out=top)
    assert np.array_equal(both, img)


def test_batching_sharding_and_bands_are_bit_invariant(cbox64):
    s, ref, c, O = cbox64
    full, st = c.render(mode="f64", spp=8)
    # many small batches (row chunks x sample batches) -> identical sums
    small, st2 = c.render(mode="f64", spp=8, batch_paths=64 * 5 * 3)
    assert st2["passes"] > 10 and np.array_equal(full, small)
    # sample-range sharding, accumulated in order -> identical; summed separately -> 1 ulp
    acc, _ = c.render(mode="f64", spp=4, sample_offset=0, spp_total=8)
    acc, _ = c.render(mode="f64", spp=4, sample_offset=4, spp_total=8, accumulate=True, out=acc)
    assert np.array_equal(acc, full)
    lo, _ = c.render(mode="f64", spp=4, sample_offset=0, spp_total=8)
    hi, _ = c.render(mode="f64", spp=4, sample_offset=4, spp_total=8)
    assert np.allclose(lo + hi, full, rtol=0, atol=1e-14)
    # interleaved row bands (the multi-GPU tile sharding): disjoint supports, exact union
    parts = [c.render(mode="f64", spp=8, row_stride=3, row_phase=k, band=4)[0] for k in range(3)]
    assert np.array_equal(parts[0] + parts[1] + parts[2], full)
    rows = np.arange(64)
    for k in range(3):
        assert (parts[k][(rows // 4) % 3 != k] == 0).all()


def test_tracer_scheduling_options_are_bit_invariant(root):
    """How paths are handed to lanes (pool chunk, grid size), whether visibility tests are parked
    in the LDS queue or run in place, and which intersector runs must not change a single bit: a
    path's result depends on its (pixel, sample) only."""
    s, ref = load(root, "config/cbox.conf", width=96, height=96, spp=6)
    base, st = make_ctx(s).render(mode="f64")
    for opts in (dict(chunk_paths=4096), dict(trace_blocks_per_cu=1), dict(shadow_queue=0),
                 dict(shadow_queue=0, brute_force_limit=0), dict(chunk_paths=64, trace_blocks_per_cu=3)):
        c = pt.Context(0, brute_force_limit=opts.pop("brute_force_limit", None))
        for k, v in opts.items():
            c.set_option(k, v)
        c.build_accel(s)
        img, st2 = c.render(mode="f64")
        assert np.array_equal(img, base), opts
        assert (st2["segments"], st2["shadow_rays"], st2["records"]) == (st["segments"], st["shadow_rays"], st["records"])
    # f32: same statement
    b32, _ = make_ctx(s).render(mode="f32")
    c = pt.Context(0)
    c.set_option("shadow_queue", 0)
    c.build_accel(s)
    assert np.array_equal(c.render(mode="f32")[0], b32)


@pytest.mark.parametrize("conf,kw", [("config/mis.conf", dict(width=96, height=64, spp=4)),
                                     ("config/cbox.conf", dict(width=64, height=64, spp=4))])
def test_bvh_stack_spill_is_bit_invariant(root, conf, kw):
    """BVH traversal stacks: however few levels are kept in LDS (deeper ones go to a global column
    per lane), closest hits and visibility tests are the same to the bit, in both precisions."""
    s, ref = load(root, conf, **kw)
    imgs = {}
    for levels in (0, 1, 3, 8):
        c = pt.Context(0, brute_force_limit=0)                      # the BVH also on the Cornell box
        c.set_option("lds_stack_levels", levels)
        c.build_accel(s)
        assert not c.accel_info()["brute_force"]
        for mode in ("f64", "f32"):
            img, st = c.render(mode=mode)
            key = (mode,)
            if key in imgs:
                assert np.array_equal(img, imgs[key][0]), (levels, mode)
                assert st["bvh_nodes"] == imgs[key][1]["bvh_nodes"] and st["bvh_tris"] == imgs[key][1]["bvh_tris"]
            else:
                imgs[key] = (img, st)
                assert st["bvh_nodes"] > 0 and st["bvh_tris"] > 0
    # and the all-triangles intersector agrees with all of them on the Cornell box
    if conf.endswith("cbox.conf"):
        base, _ = make_ctx(s).render(mode="f64")
        assert np.array_equal(base, imgs[("f64",)][0])


@pytest.mark.parametrize("conf,kw", [("config/mis.conf", dict(width=96, height=64, spp=4)),
                                     ("config/room.conf", dict(width=96, height=64, spp=2)),
                                     ("config/cbox.conf", dict(width=64, height=64, spp=4))])
def test_reference_splitting_is_bit_invariant(root, conf, kw):
    """Option bvh_split: triangles with mostly empty boxes hang in the tree through several references
    (each with the box of its part).  The intersector tests the whole triangle wherever it meets it, so
    first hits, hit points and images are those of the unsplit tree to the bit, in both precisions and
    under both estimators; only the visit counts move (fewer triangle tests)."""
    s, ref = load(root, conf, **kw)
    out = {}
    for split in (0, 30, 200):
        c = pt.Context(0, brute_force_limit=0)
        c.set_option("bvh_split", split)
        c.build_accel(s)
        assert (c.accel_info()["n_leaves"] > ref.verts.shape[0]) == (split > 0)
        ids, pts = c.first_hit(with_points=True)
        got = [ids, pts]
        for mode, est in (("f64", "reference"), ("f32", "reference"), ("f64", "fast")):
            img, st = c.render(mode=mode, estimator=est)
            got += [img, np.array([st["segments"], st["shadow_rays"], st["records"]])]
        if split == 0:
            out = got
        else:
            for a, b in zip(out, got):
                assert np.array_equal(a, b, equal_nan=True), split


@pytest.mark.parametrize("conf,kw", [("config/mis.conf", dict(width=160, height=96, spp=8)),
                                     ("config/room.conf", dict(width=96, height=64, spp=3)),
                                     ("config/cbox.conf", dict(width=96, height=96, spp=6))])
def test_ring_tracer_equals_lane_tracer(root, conf, kw):
    """BVH scenes with at most one square light (the Cornell box forced onto the BVH: its visibility
    tests travel through the rings too): the tracer that queues hits and rays per wave (more paths
    in flight than lanes), the one that keeps every path on its lane and the wavefront tracer (the
    Cornell box's visibility tests through its test queue) render the same bits, run the same visibility tests and visit
    the same BVH records, whatever the grid, the batches and the LDS share of the stacks and nodes."""
    s, ref = load(root, conf, **kw)
    out = {}
    for rings, extra in ((0, {}), (1, dict(bvh_wavefront=0)), (1, dict(bvh_wavefront=0, trace_blocks_per_cu=1, lds_stack_levels=4)),
                         (1, dict(bvh_wavefront=0, chunk_paths=4096, lds_nodes=0)),
                         # the wavefront tracer (traversal and shading in separate kernels): default options, one workgroup
                         # per CU with 4 stack levels in LDS, no nodes in LDS and whole-wave refills, and batches of 1000 paths
                         (1, dict(bvh_wavefront=1)), (1, dict(bvh_wavefront=1, trace_blocks_per_cu=1, lds_stack_levels=4)),
                         (1, dict(bvh_wavefront=1, lds_nodes=0, wf_refill=64)), (1, dict(bvh_wavefront=1, default_batch_paths=1000, wf_refill=1))):
        c = pt.Context(0, brute_force_limit=0)
        c.set_option("bvh_rings", rings)
        for k, v in extra.items():
            c.set_option(k, v)
        c.build_accel(s)
        assert not c.accel_info()["brute_force"]
        for mode in ("f64", "f32"):
            img, st = c.render(mode=mode)
            assert ("k_wf_" in c.last_kernel()) == bool(rings and extra.get("bvh_wavefront")), c.last_kernel()
            if mode in out:
                assert np.array_equal(img, out[mode][0]), (rings, extra, mode)
                wf = "k_wf_" in c.last_kernel()
                for k in ("segments", "records", "shadow_rays", "paths") + (() if wf else ("bvh_nodes", "bvh_tris")):
                    assert st[k] == out[mode][1][k], (k, rings, extra, mode)
                if wf:
                    # the samples of a pixel share their camera ray (no per-sample jitter in the reference): the
                    # wavefront tracer traverses it once per pixel and batch, the others once per sample
                    # (batches that hold one sample per pixel share nothing)
                    small = "default_batch_paths" in extra
                    assert 0 < st["bvh_nodes"] <= out[mode][1]["bvh_nodes"] and 0 < st["bvh_tris"] <= out[mode][1]["bvh_tris"]
                    assert small or st["bvh_nodes"] < out[mode][1]["bvh_nodes"]
            else:
                out[mode] = (img, st)


def test_render_f32_meets_rmse_bar(root):
    s, ref = load(root, "config/cbox.conf", width=64, height=64, spp=16)
    c = make_ctx(s)
    a = oracle.Oracle(ref, seed=13).render()
    b = oracle.Oracle(ref, seed=14).render()
    tol = max(2 * oracle.rmse_rel(b, a), 1e-3)
    img, st = c.render(mode="f32", seed=13)
    assert oracle.rmse_rel(img, a) <= tol
    # an independent seed must also sit inside the statistical bar
    img2, _ = c.render(mode="f32", seed=99)
    assert oracle.rmse_rel(img2, a) <= tol
    # and the f32 path length statistics stay those of the reference estimator
    _, ost = oracle.Oracle(ref, seed=13).render(stats=True)
    assert abs(st["segments"] / ost["segments"] - 1) < 0.01


def test_full_size_properties(root):
    """Headline size (1024 x 1024): properties that need no CPU render."""
    s, ref = load(root, "config/cbox.conf", width=1024, height=1024, spp=4)
    c = make_ctx(s)
    a, sa = c.render(mode="f64", spp=4)
    b, sb = c.render(mode="f64", spp=4, batch_paths=1 << 19)
    assert np.array_equal(a, b) and sa["segments"] == sb["segments"] and sb["passes"] > sa["passes"]
    ids = c.first_hit()
    lit = ((ids == 10) | (ids == 11)).reshape(1024, 1024)
    assert np.array_equal(a[lit], np.full((lit.sum(), 3), 10.0))       # emitter seen directly: Ka, unclamped
    assert a[~lit].max() <= 1.0 + 1e-12                                 # every other sample is clamped to 1
    assert (a[(ids == -1).reshape(1024, 1024)] == 0).all()              # misses are black
    fin = pt.finalize_image(a)
    assert fin.max() <= 1.0 and fin.min() >= 0
    f32, _ = c.render(mode="f32", spp=4)
    assert oracle.rmse_rel(f32, a) < 1e-3
    # a crop of the full-size image against the oracle rendering only that window
    ref.region = (500, 480, 515, 543)
    O = oracle.Oracle(ref, seed=13)
    oimg = O.render()
    assert_f64_parity(a[500:516, 480:544], oimg[500:516, 480:544], ctx=c, O=O, origin=(500, 480))


def test_full_size_properties_bvh_scene(root, room_scene):
    """room (336 k triangles, BVH, ring tracer) at 1024 x 1024: cuts of the frame add up to the
    frame, and a window of it equals the oracle rendering only that window."""
    s, ref = load(root, "config/room.conf", width=1024, height=1024, spp=2)
    ref.use_bvh = True
    c = make_ctx(s)
    assert not c.accel_info()["brute_force"]
    a, sa = c.render(mode="f64", spp=2)
    b, sb = c.render(mode="f64", spp=2, batch_paths=1 << 19)
    assert np.array_equal(a, b) and sa["segments"] == sb["segments"] and sb["passes"] > sa["passes"]
    bands = [c.render(mode="f64", spp=2, row_stride=2, row_phase=k, band=8)[0] for k in range(2)]
    assert np.array_equal(bands[0] + bands[1], a)                       # disjoint rows: exact union
    assert (bands[0][8:16] == 0).all() and (bands[1][0:8] == 0).all()
    halves = [c.render(mode="f64", spp=1, sample_offset=k, spp_total=2)[0] for k in range(2)]
    assert np.abs(halves[0] + halves[1] - a).max() <= 1e-15 * max(1.0, a.max())
    ids = c.first_hit()
    assert (a[(ids == -1).reshape(1024, 1024)] == 0).all()
    ref.region = (600, 400, 607, 527)
    O = oracle.Oracle(ref, seed=13)
    oimg = O.render()
    assert_f64_parity(a[600:608, 400:528], oimg[600:608, 400:528], depth=5, ctx=c, O=O, origin=(600, 400))
    assert np.array_equal(ids.reshape(1024, 1024)[600:608, 400:528], O.first_hit().reshape(1024, 1024)[600:608, 400:528])


def test_pathtracer_process_writes_reference_format_ppm(root, tmp_path):
    import json
    conf = json.load(open(os.path.join(root, "config/cbox.conf")))
    conf["MainWindowInfo"]["Width"] = conf["MainWindowInfo"]["Height"] = 24
    conf["PathTracer"]["DrawRegion"] = [0, 0, 24, 24]
    conf["PathTracer"]["Samples"] = 4
    conf["PathTracer"]["ResultPath"] = str(tmp_path / "out.ppm")
    p = tmp_path / "c.conf"
    p.write_text(json.dumps(conf))
    tracer = pt.PathTracer(str(p), root_dir=root)
    with pytest.raises(pt.PtxError):
        tracer.Process()                                   # Process before Init is an error
    tracer.Init()
    img = tracer.Process()
    ref = scene_io.load_conf(str(p), root_dir=root)
    ofin = oracle.finalize(oracle.Oracle(ref, seed=13).render())
    got = pt.read_ppm(str(tmp_path / "out.ppm"))
    want = (ofin * 255).astype(np.int64)
    assert got.shape == (24, 24, 3)
    assert (np.abs(got - want) <= 1).all() and (got != want).mean() < 0.01   # int(c*255) truncation
    text = open(tmp_path / "out.ppm").read()
    assert text.startswith("P3\n24 24\n255\n") and text.count("\n") == 3


def test_errors_cross_the_boundary_as_codes(root):
    c = pt.Context(0)
    with pytest.raises(pt.PtxError) as e:
        c.width = c.height = 4
        c.first_hit()
    assert e.value.code == 5                                # no scene yet
    with pytest.raises(pt.PtxError) as e:
        pt.Context(99)
    assert e.value.code == 1
