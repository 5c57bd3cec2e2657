"""CPU tests of the oracle: golden vectors, and the properties the reference's author probed in
the commented-out checks of cPathTracer::GetDrawResources (src/tracing/cPathTracer.cpp:384-511 of
the reference): barycentric round trip, sampling sanity, refraction lobes."""
import os

import numpy as np
import pytest

from oracle import oracle, scene_io


def test_tri_known_answers(golden):
    g = np.load(os.path.join(golden, "tri_kat.npz"))
    for i in range(len(g["o"])):
        ok, p = oracle.tri_raycast(g["o"][i], g["d"][i], g["p1"][i], g["p2"][i], g["p3"][i])
        assert ok == bool(g["accept"][i])
        assert np.array_equal(p, g["pt"][i], equal_nan=True)
        t = oracle.tri_raycast_t(g["o"][i], g["d"][i], g["p1"][i], g["p2"][i], g["p3"][i])
        assert (np.isnan(t) and np.isnan(g["t"][i])) or t == g["t"][i]


def test_tri_slack_semantics():
    tri = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    d = (0, 0, -1)
    # barycentric slack is 1e-10 on both sides (cMathUtil.hpp:84-89 default eps)
    assert oracle.tri_raycast((-5e-11, 0.3, 1), d, *tri)[0]
    assert not oracle.tri_raycast((-2e-10, 0.3, 1), d, *tri)[0]
    # t > -1e-10
    assert oracle.tri_raycast((0.2, 0.2, -5e-11), d, *tri)[0]
    assert not oracle.tri_raycast((0.2, 0.2, -2e-10), d, *tri)[0]
    # singular system -> reject
    assert not oracle.tri_raycast((0.2, 0.2, 1), (1, 0, 0), *tri)[0]
    assert np.isnan(oracle.tri_raycast_t((0.2, 0.2, 1), (1, 0, 0), *tri))


def test_barycentric_round_trip():
    # the reference's own check: weights recovered from a convex combination (cPathTracer.cpp:480-511)
    rng = np.random.default_rng(3)
    for _ in range(200):
        v = rng.uniform(-1, 1, (3, 3))
        w = rng.random(3)
        w /= w.sum()
        p = w @ v
        b = oracle.barycentric(p, v[0], v[1], v[2])
        assert np.allclose(b / b.sum(), w, atol=1e-8)
    # a plane through the origin along an axis (the Cornell box floor, y = 0) is regularised
    a, b_, c = (5.5, 0, 0), (0, 0, 0), (0, 0, 5.6)
    w = np.array([0.2, 0.5, 0.3])
    p = w @ np.array([a, b_, c])
    got = oracle.barycentric(p, a, b_, c)
    assert np.allclose(got / got.sum(), w, atol=1e-5)


def test_reflect_refract_identities():
    rng = np.random.default_rng(5)
    for _ in range(100):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        i = rng.normal(size=3); i /= np.linalg.norm(i)
        r = oracle.reflect(n, i)
        assert abs(np.linalg.norm(r) - 1) < 1e-12
        assert abs(r @ n + i @ n) < 1e-12               # mirror about the tangent plane
        t = oracle.refract(n, i, 1 / 1.5)
        if not np.isnan(t).any():
            assert abs(np.linalg.norm(t) - 1) < 1e-9
            # Snell: sin(theta_t) = eta * sin(theta_i) with eta flipped when leaving the medium
            eta = 1 / 1.5 if n @ i <= 0 else 1.5
            assert abs(np.linalg.norm(np.cross(n, t)) - eta * np.linalg.norm(np.cross(n, i))) < 1e-9
            assert (t @ n) * (i @ n) > 0                # continues through the surface
    # eta = 1 passes straight through
    assert np.allclose(oracle.refract((0, 1, 0), (0.6, -0.8, 0), 1.0), (0.6, -0.8, 0))
    # total internal reflection gives NaN
    assert np.isnan(oracle.refract((0, 1, 0), (0.9, 0.43588989435, 0), 1 / 1.5)).any()


def test_hemisphere_sampler_is_the_references_pole_peaked_one():
    n = np.array([0.0, 1.0, 0.0])
    rng = np.random.default_rng(9)
    xs = rng.random((4000, 2))
    d = np.array([oracle.sample_hemisphere(n, a, b) for a, b in xs])
    assert np.allclose(np.linalg.norm(d, axis=1), 1, atol=1e-12)
    assert (d[:, 1] >= -1e-12).all()
    # height = sqrt(1 - (1 - xi1)^2): mean pi/4, not the 1/2 of a uniform hemisphere
    assert abs(d[:, 1].mean() - np.pi / 4) < 0.02
    # rotated frame: all samples stay on the side of the normal
    for nn in [(1, 0, 0), (0, -1, 0), (0.3, -0.5, 0.81)]:
        nn = np.array(nn) / np.linalg.norm(nn)
        dd = np.array([oracle.sample_hemisphere(nn, a, b) for a, b in xs[:500]])
        assert (dd @ nn >= -1e-9).all()
        assert abs((dd @ nn).mean() - np.pi / 4) < 0.05


def test_sample_tables(golden):
    g = np.load(os.path.join(golden, "samples.npz"))
    for ni, n in enumerate(g["normals"]):
        for xi_i, (a, b) in enumerate(g["xi"]):
            assert np.array_equal(oracle.sample_hemisphere(n, a, b), g["hemi"][ni, xi_i])
    k = 0
    for r in g["refs"]:
        for a, b in g["xi"][:4]:
            pdf, ro, rd = oracle.square_light_sample(g["square_light"], r, a, b)
            assert np.array_equal(np.concatenate([[pdf], ro, rd]), g["square_out"][k])
            pdf, ro, rd = oracle.sphere_light_sample(g["sphere_light"], r, a, b, 0.5 * (a + b))
            assert np.array_equal(np.concatenate([[pdf], ro, rd]), g["sphere_out"][k])
            k += 1
    for row in g["bxdf_eval"]:
        m, n, wi, wo, f = row[:12], row[12:15], row[15:18], row[18:21], row[21:24]
        assert np.array_equal(oracle.bxdf_eval(m, wi, wo, n), f, equal_nan=True)
    for row in g["bxdf_sample"]:
        m, n, wo, x, wi, coef = row[:12], row[12:15], row[15:18], row[18:20], row[20:23], row[23:26]
        w2, c2 = oracle.bxdf_sample(m, n, wo, x)
        assert np.array_equal(w2, wi, equal_nan=True) and np.array_equal(c2, coef, equal_nan=True)


def test_lights_semantics(golden):
    g = np.load(os.path.join(golden, "samples.npz"))
    sq, sp = g["square_light"], g["sphere_light"]
    # one-sided: a point above the downward-facing quad gets pdf 0
    assert oracle.square_light_sample(sq, (2.8, 5.6, 2.8), 0.3, 0.7)[0] == 0
    # below: pdf = r^2 / (A cos) / (2 pi)  (the reference's extra 1/(2 pi), cLight.cpp:101-102)
    pdf, ro, rd = oracle.square_light_sample(sq, (2.8, 1.0, 2.8), 0.5, 0.5)
    r2 = np.sum((ro - np.array([2.8, 1.0, 2.8])) ** 2)
    area = 1.30 * 1.05
    assert np.isclose(pdf, r2 / (area * (rd @ np.array([0, -1.0, 0]))) / (2 * np.pi), rtol=1e-12)
    assert np.allclose(ro, (2.78, 5.48, 2.795))
    # sphere light outside: ray starts AT the shaded point, direction centre -> point, cone pdf
    ref = np.array([0.95, 1.6, -2.0])
    pdf, ro, rd = oracle.sphere_light_sample(sp, ref, 0.1, 0.2, 0.3)
    assert np.array_equal(ro, ref) and np.allclose(rd, (0, -1, 0))
    cos_max = np.cos(np.arcsin(0.35 / 2.0))
    assert np.isclose(pdf, 1 / (2 * np.pi * (1 - cos_max)), rtol=1e-12)
    # inside: uniform sphere pdf
    assert np.isclose(oracle.sphere_light_sample(sp, (0.95, 3.6, -2.0), 0.1, 0.2, 0.3)[0], 1 / (4 * np.pi))


def test_bxdf_semantics():
    diffuse = np.zeros(12); diffuse[3:6] = (0.85, 0.75, 0.65); diffuse[9] = 1; diffuse[10] = 1
    glossy = np.zeros(12); glossy[3:6] = 0.77; glossy[6:9] = 0.97; glossy[9] = 1; glossy[10] = 98
    glass = np.zeros(12); glass[9] = 1.5; glass[10] = 1
    n = np.array([0, 1.0, 0])
    wi = np.array([0.6, -0.8, 0])            # light -> point
    wo = np.array([-0.6, 0.8, 0])
    assert np.allclose(oracle.bxdf_eval(diffuse, wi, wo, n), diffuse[3:6] / np.pi)
    mirror = oracle.reflect(n, wi)
    f = oracle.bxdf_eval(glossy, wi, mirror, n)
    assert np.allclose(f, 0.77 / np.pi + 0.97 * 100 / (2 * np.pi))
    # even exponent: the lobe is mirrored for negative cosines (cos_alpha not clamped)
    assert np.allclose(oracle.bxdf_eval(glossy, wi, -mirror, n), f)
    # glass: class chosen by |Ni - 1| >= 1e-5, coefficient (1,1,1), Schlick split at xi < R
    R0 = ((1 - 1.5) / (1 + 1.5)) ** 2
    R = R0 + (1 - R0) * (1 - 0.8) ** 5
    w_r, c_r = oracle.bxdf_sample(glass, n, wo, [R * 0.999, 0])
    w_t, c_t = oracle.bxdf_sample(glass, n, wo, [R * 1.001, 0])
    assert np.allclose(c_r, 1) and np.allclose(c_t, 1)
    assert np.allclose(w_r, oracle.reflect(n, -wo)) and w_t[1] < 0
    assert np.allclose(w_t, oracle.refract(n, -wo, 1 / 1.5))
    # NEE eval of glass: 1/(4 pi) max(0, ideal . wo)
    ideal = oracle.refract(n, wi, 1 / 1.5)
    assert np.allclose(oracle.bxdf_eval(glass, wi, ideal, n), 1 / (4 * np.pi))
    assert np.allclose(oracle.bxdf_eval(glass, wi, -ideal, n), 0)


def test_rng_is_uniform_and_keyed():
    a = np.array([oracle.rng(13, p, s, b, d) for p in range(40) for s in range(5) for b in range(5) for d in range(5)])
    assert a.min() >= 0 and a.max() < 1
    assert abs(a.mean() - 0.5) < 0.02 and abs(a.var() - 1 / 12) < 0.01
    assert len(np.unique(a)) == len(a)
    assert oracle.rng(13, 1, 2, 3, 4) == oracle.rng(13, 1, 2, 3, 4)
    assert oracle.rng(13, 1, 2, 3, 4) != oracle.rng(14, 1, 2, 3, 4)


@pytest.fixture(scope="module")
def cbox(root):
    return scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=32, height=32, spp=16)


def test_first_hit_golden(root, golden):
    g = np.load(os.path.join(golden, "cbox_first_hit.npz"))
    sc = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=64, height=64, spp=1)
    ids = oracle.Oracle(sc).first_hit()
    assert np.array_equal(ids, g["ids_64"].astype(np.int32))
    # the box is framed by misses (-1); the light and the cube (faces 14..25) are visible
    assert ids.min() == -1 and {10, 11} <= set(ids.tolist()) and set(range(14, 26)) & set(ids.tolist())


def test_closest_hit_ties_keep_lowest_face_id(cbox):
    O = oracle.Oracle(cbox)
    # straight up through the middle of the light: the luminaire (faces 10,11, y=5.487) is hit
    # before the coplanar ceiling/hole quads (y=5.488); a ray along the quad diagonal of the
    # floor hits faces 6 and 7 at the same distance and must report 6
    f, p = O.closest_hit((2.78, 1.0, 2.795), (0, 1, 0))
    assert f in (10, 11) and abs(p[1] - 5.487) < 1e-6
    v = cbox.verts
    a, c = v[6, 0], v[6, 2]                    # shared diagonal of the floor quad (faces 6 and 7)
    mid = 0.5 * (a + c)
    f, p = O.closest_hit(mid + np.array([0, 2.0, 0]), (0, -1, 0))
    assert f in (6, 7)                         # a near-tie: decided by last-ulp rounding
    # an exact tie (the same triangle twice) keeps the lowest id: strict '<' in ascending order
    import copy
    dup = copy.copy(cbox)
    dup.verts = np.concatenate([cbox.verts[6:7], cbox.verts[6:7], cbox.verts])
    dup.normals = np.concatenate([cbox.normals[6:7], cbox.normals[6:7], cbox.normals])
    dup.mat_ids = np.concatenate([cbox.mat_ids[6:7], cbox.mat_ids[6:7], cbox.mat_ids])
    f, _ = oracle.Oracle(dup).closest_hit((1.0, 2.0, 1.0), (0, -1, 0))
    assert f == 0
    # self-intersection guard: hits closer than 1e-6 are ignored
    f, _ = O.closest_hit((2.0, 5e-7, 2.0), (0, -1, 0))
    assert f != 6 and f != 7


def test_tie_alternatives_name_every_way_a_path_can_fall(root):
    """The tie choice behind the f64 parity rule (oracle/parity.py): mask 0 is the reference's own
    path; bit k hands the k-th tie of the path to the runner-up, which changes the face reported
    there and everything after it; a path without ties has one alternative -- itself."""
    from oracle import parity
    sc = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=64, height=64, spp=8)
    O = oracle.Oracle(sc, seed=13)
    # (column, row, sample) of paths that cross the coplanar ceiling quads, found by scanning the frame once
    for x, y, s in ((16, 18, 5), (14, 29, 1), (51, 49, 7)):
        rgb0, faces0, margins0, nt = O.trace_sample_ties(x, y, s, 0)
        plain, pfaces, pmargins = O.trace_sample(x, y, s)
        assert np.array_equal(rgb0, plain) and np.array_equal(faces0, pfaces)          # mask 0 = the reference's rule
        assert nt == 1 and margins0.min() <= parity.TIE_GAP
        k = int(np.argmin(margins0))
        rgb1, faces1, _, nt1 = O.trace_sample_ties(x, y, s, 1)
        assert np.array_equal(faces1[:k], faces0[:k]) and faces1[k] != faces0[k]       # the other face at the tie
        assert np.abs(rgb1 - rgb0).max() > 1e-6                                        # and another sample value
        alts = O.tie_alternatives(x, y, s)
        assert sorted(m for m, _, _ in alts)[:2] == [0, 1] and len(alts) == 1 + max(1, nt1)
        # the matching rule: the oracle's other path is accepted as a tie, its own as equal, anything else is not
        assert parity.match_sample(rgb0, O, x, y, s)[0] == "equal"
        assert parity.match_sample(rgb1, O, x, y, s)[:2] == ("tie", 1)
        assert parity.match_sample(0.5 * (rgb0 + rgb1), O, x, y, s)[0] == "unmatched"
    rgb, faces, margins, nt = O.trace_sample_ties(30, 40, 0, 0)
    assert nt == 0 and len(O.tie_alternatives(30, 40, 0)) == 1
    assert parity.match_sample(rgb + 1e-6, O, 30, 40, 0)[0] == "unmatched"             # no tie: nothing excuses a difference
    # a bit beyond the ties a path meets changes nothing
    assert np.array_equal(O.trace_sample_ties(30, 40, 0, 0xff)[0], rgb)
    # the oracle's culling tree reports the same ties as its plain loop
    Ob = oracle.Oracle(sc, seed=13, use_bvh=True)
    for m in (0, 1):
        a, b = O.trace_sample_ties(16, 18, 5, m), Ob.trace_sample_ties(16, 18, 5, m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3]


def test_vistest_semantics(cbox):
    O = oracle.Oracle(cbox)
    light_pt = (2.78, 5.48, 2.795)
    assert O.vistest(light_pt, (0.5, 0.0, 0.5))            # floor corner sees the light
    assert not O.vistest(light_pt, (3.3, 0.0, 1.6))        # floor under the cube is shadowed
    assert O.vistest((1, 1, 1), (1, 1, 1))                 # zero-length segment: NaN direction -> visible
    # the end point's own surface does not block (t < dist - 1e-6)
    assert O.vistest(light_pt, (1.0, 0.0, 4.0))


def test_render_golden_and_determinism(cbox, golden):
    g = np.load(os.path.join(golden, "cbox_32_spp16.npz"))
    img, st = oracle.Oracle(cbox, seed=13).render(stats=True)
    assert np.array_equal(img, g["img"])
    assert [st["paths"], st["segments"], st["shadow_rays"]] == g["stats"].tolist()
    img1 = oracle.Oracle(cbox, seed=13).render(threads=1)
    assert np.array_equal(img, img1)                        # thread count does not change the image


def test_sample_range_sharding_is_linear(cbox):
    O = oracle.Oracle(cbox, seed=13)
    full = O.render(spp=16)
    lo = O.render(spp=8, sample_offset=0, spp_total=16)
    hi = O.render(spp=8, sample_offset=8, spp_total=16)
    assert np.allclose(lo + hi, full, rtol=0, atol=1e-14)


def test_estimator_properties(cbox):
    O = oracle.Oracle(cbox, seed=13)
    img = O.render(spp=16)
    fin = oracle.finalize(img)
    assert fin.min() >= 0 and fin.max() <= 1.0
    # per-sample values are clamped to max 1 except paths that END on the emitter at bounce 0
    ids = O.first_hit()
    lit = (ids == 10) | (ids == 11)
    assert np.allclose(img.reshape(-1, 3)[lit], 10.0)
    assert img.reshape(-1, 3)[~lit].max() <= 1.0 + 1e-12
    # red wall on the left of the image, blue on the right (camera looks down +z from z = -8)
    assert fin[16, 1, 0] > 4 * fin[16, 1, 2] and fin[16, 30, 2] > 4 * fin[16, 30, 0]


def test_region_and_flags(root):
    sc = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=24, height=24, spp=2)
    sc.region = (4, 6, 10, 12)
    img = oracle.Oracle(sc).render()
    mask = np.zeros((24, 24), bool)
    mask[4:11, 6:13] = True                                  # inclusive ends
    assert (img[~mask] == 0).all() and (img[mask].sum(axis=-1) > 0).any()
    sc.region = (0, 0, 24, 24)
    sc.enable_indirect = False
    img_d, st = oracle.Oracle(sc).render(stats=True)
    assert st["segments"] == st["paths"]                     # one bounce only
    sc.enable_indirect = True
    sc.enable_bary_normal = False
    img_f = oracle.Oracle(sc).render()
    sc.enable_bary_normal = True
    img_b = oracle.Oracle(sc).render()
    # the Cornell box has flat per-face normals equal to the geometric ones up to sign; the red
    # wall's vertex normal is slightly tilted, so images agree closely but not exactly
    assert oracle.rmse_rel(img_f, img_b) < 0.5


def test_bvh_oracle_equals_brute_force(root):
    sc = scene_io.load_conf(os.path.join(root, "config/mis.conf"), root_dir=root, width=48, height=32, spp=2)
    a = oracle.Oracle(sc, use_bvh=False)
    b = oracle.Oracle(sc, use_bvh=True)
    ia, pa = a.first_hit(with_points=True)
    ib, pb = b.first_hit(with_points=True)
    assert np.array_equal(ia, ib) and np.array_equal(pa, pb, equal_nan=True)
    assert np.array_equal(a.render(), b.render())


def test_fast_estimator_strategies_agree_in_the_mean(root):
    """The physically based estimator of the oracle: light sampling alone, BRDF sampling alone and
    their balance-heuristic combination are three estimators of the same integral wherever no delta
    lobe is involved -- which pins the light densities (square: area -> solid angle; sphere: cone)
    against the BRDF's (cosine + Phong lobe) and against the ray / light intersections."""
    for conf, w, h in (("config/mis.conf", 16, 12), ("config/cbox_glossy.conf", 16, 16)):
        ref = scene_io.load_conf(os.path.join(root, conf), root_dir=root, width=w, height=h, spp=4096, depth=4)
        ref.mats = ref.mats.copy()
        ref.mats[:, 9] = 1.0                                     # no glass: every lobe has a density
        m = [oracle.Oracle(ref, seed=13, estimator="fast", fast_strategy=k).render().mean(axis=(0, 1)) for k in (0, 1, 2)]
        assert np.abs(m[1] / m[0] - 1).max() < 0.02, conf
        # BRDF sampling alone rarely finds the small, bright sphere lights of mis: a noisy mean
        assert np.abs(m[2] / m[0] - 1).max() < (0.1 if "mis" in conf else 0.03), conf
        # and it is not the reference's estimator (whose light sampling is 2 pi too bright in the Cornell box)
        if "cbox" in conf:
            assert oracle.Oracle(ref, seed=13).render().mean() > 1.5 * m[0].mean()


def test_sum_order_variants_agree_on_what_is_compared_bit_for_bit(root):
    """The restatement in the three plausible orders of Eigen's 3- and 4-term sums: first-hit face
    ids are the same on every pixel, hit points move in their last bits only, images by far less
    than the parity bar (profiles/r02_sum_order_report.json counts the same at 1024 x 1024)."""
    out = {}
    try:
        for v in (0, 1, 2):
            oracle.use_sum_order(v)
            assert oracle.lib().orc_sum_order() == v
            ref = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=96, height=96, spp=2)
            O = oracle.Oracle(ref, seed=13)
            ids, pts = O.first_hit(with_points=True)
            out[v] = (ids, pts, O.render())
    finally:
        oracle.use_sum_order(0)
    for v in (1, 2):
        assert np.array_equal(out[v][0], out[0][0])
        assert np.nanmax(np.abs(out[v][1] - out[0][1])) < 1e-13
        assert (out[v][1] != out[0][1]).any()                       # they ARE different arithmetic
        assert np.abs(out[v][2] - out[0][2]).max() < 1e-9 or (np.abs(out[v][2] - out[0][2]).max(axis=-1) > 1e-9).sum() <= 2


def test_amplification_names_the_paths_that_multiply_their_roundings(root):
    """The second exception of the f64 parity rule (oracle/parity.py): the oracle measures, on the sample
    itself, by how much the path multiplies an error of one of its directions.  Flat, diffuse geometry
    does not (the Cornell box: far below the 4 400 at which 1024 ulps reach the 1e-9 bar, so nothing
    there is ever excused); a path over the smoothly shaded furniture of room.obj does -- a hundredfold
    per bounce -- and only there may a difference beyond the bar pass, up to 1024 ulps times the factor."""
    from oracle import parity
    sc = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=64, height=64, spp=8)
    O = oracle.Oracle(sc, seed=13)
    amps = [O.amplification(x, y, 0) for x, y in ((30, 40), (10, 50), (50, 12), (32, 32))]
    assert all(0.0 <= a < 1000.0 for a in amps), amps
    rgb = O.trace_sample(30, 40, 0)[0]
    assert parity.match_sample(rgb + 2e-9, O, 30, 40, 0)[0] == "unmatched"
    room = scene_io.load_conf(os.path.join(root, "config/room.conf"), root_dir=root, width=3840, height=2160, spp=1)
    R = oracle.Oracle(room, seed=13)
    x, y = 210, 567                          # a five-bounce path over curved furniture (found by tools/prove_frame.py)
    amp = R.amplification(x, y, 0)
    assert 5e4 < amp < 5e5, amp
    assert abs(R.amplification(x, y, 0, eps=1e-11) / amp - 1.0) < 0.05   # a derivative, not an artefact of the step
    rgb = R.trace_sample(x, y, 0)[0]
    status, _, detail = parity.match_sample(rgb + 2.8e-9, R, x, y, 0)    # what the GPU differs by on this sample
    assert status == "amplified" and detail["amplification"] == amp
    assert parity.match_sample(rgb + 1e-6, R, x, y, 0)[0] == "unmatched"  # 40 times the allowance: a bug, not a rounding
    assert R.amplification(100, 100, 0) < 100.0                          # a wall: the same scene, an ordinary path
    # the physically based estimator has no such probe: nothing is excused there
    F = oracle.Oracle(sc, seed=13, estimator="fast")
    assert F.amplification(30, 40, 0) is None
