"""The physically based estimator (PTX_ESTIMATOR_FAST: balance-heuristic MIS of light and BRDF
sampling, lobe importance sampling, Russian roulette, sphere lights with a visibility test) on the
GPU against the oracle running the SAME estimator on the same random numbers -- never against the
reference's estimator, from which it differs by design."""
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from oracle import oracle, scene_io
from test_gpu_parity import assert_f64_parity, load, make_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conf,w,h,spp,kw", [
    ("config/cbox.conf", 64, 64, 8, {}),                       # one square light: tests parked, glass cube
    ("config/cbox.conf", 48, 48, 6, dict(shadow_queue=0)),     # the same with tests in place
    ("config/cbox.conf", 48, 48, 6, dict(brute_force_limit=0)),  # through the BVH
    ("config/cbox_glossy.conf", 48, 48, 8, {}),                # Phong lobe Ns = 98
    ("config/cbox_cube.conf", 48, 48, 8, {}),
    ("config/mis.conf", 96, 64, 4, {}),                        # five sphere lights, each with its visibility test: wavefront tracer
    ("config/mis.conf", 96, 64, 4, dict(bvh_wavefront=0)),     # the same on the ring tracer
    ("config/mis.conf", 96, 64, 4, dict(bvh_rings=0)),         # and on the lane tracer
    ("config/room.conf", 96, 64, 2, {}),
    ("config/room.conf", 96, 64, 2, dict(bvh_wavefront=0)),
    ("config/room.conf", 64, 48, 2, dict(bvh_rings=0))])
def test_fast_estimator_f64_matches_oracle(root, conf, w, h, spp, kw):
    s, ref = load(root, conf, width=w, height=h, spp=spp)
    c = pt.Context(0, brute_force_limit=kw.get("brute_force_limit"))
    for k in ("shadow_queue", "bvh_rings", "bvh_wavefront"):
        if k in kw:
            c.set_option(k, kw[k])
    c.build_accel(s)
    O = oracle.Oracle(ref, seed=13, estimator="fast")
    img, st = c.render(mode="f64", seed=13, estimator="fast")
    oimg, ost = O.render(stats=True)
    assert "fast" in c.last_kernel()
    if not c.accel_info()["brute_force"]:
        wavefront = bool(kw.get("bvh_rings", 1) and kw.get("bvh_wavefront", 1))
        assert ("k_wf_" in c.last_kernel()) == wavefront
        assert ("rings" in c.last_kernel()) == bool(kw.get("bvh_rings", 1) and not wavefront)
    assert st["paths"] == w * h * spp and st["shadow_rays"] > 0
    # relative bar: radiances reach 140 on mis (no clamp in this estimator)
    scale = np.maximum(1.0, np.abs(oimg).max(axis=-1, keepdims=True))
    assert_f64_parity(img / scale, oimg / scale, st, ost, depth=ref.depth, ctx=c, O=O, render_kw=dict(estimator="fast"))
    assert st["shadow_rays"] == ost["shadow_rays"] or abs(st["shadow_rays"] - ost["shadow_rays"]) <= 8 * ref.depth


def test_fast_and_reference_estimators_differ(root):
    """Two estimators, two images: the reference's light sampling is 2 pi too bright, counts the
    Cornell box's lamp twice and clamps every bounce."""
    s, ref = load(root, "config/cbox.conf", width=48, height=48, spp=64)
    c = make_ctx(s)
    a, sa = c.render(mode="f64", estimator="reference")
    b, sb = c.render(mode="f64", estimator="fast")
    b2, _ = c.render(mode="f64", estimator="fast", seed=77)
    noise = oracle.rmse_rel(b2, b)
    assert oracle.rmse_rel(a, b) > 20 * noise
    assert sb["segments"] < sa["segments"]                      # roulette and emitter rules shorten the paths
    assert b.mean() < a.mean()


def test_fast_estimator_ring_tracer_equals_lane_tracer(root):
    """BVH scenes: the wavefront tracer, the ring tracer and the lane tracer run the physically based
    estimator on the same paths -- the same numbers of segments, records and visibility tests -- to the parity bar.
    (Not to the bit: the shading helpers allow fused multiply-adds, the compiler takes that licence
    differently in the two kernels, and a Phong exponent of 10 000 turns an ulp into 1e-11.)"""
    for conf, kw in (("config/mis.conf", dict(width=160, height=96, spp=6)), ("config/cbox.conf", dict(width=64, height=64, spp=4))):
        s, _ = load(root, conf, **kw)
        out = {}
        for rings, wf in ((0, 0), (1, 0), (1, 1)):
            c = pt.Context(0, brute_force_limit=0)
            c.set_option("bvh_rings", rings)
            c.set_option("bvh_wavefront", wf)
            c.build_accel(s)
            for mode in ("f64", "f32"):
                img, st = c.render(mode=mode, estimator="fast")
                assert ("rings" in c.last_kernel()) == bool(rings and not wf) and ("k_wf_" in c.last_kernel()) == bool(wf)
                if mode in out:
                    tol = 1e-9 if mode == "f64" else 1e-3
                    assert np.abs(img - out[mode][0]).max() <= tol * max(1.0, float(np.abs(img).max())), (conf, mode)
                    for k in ("segments", "records", "shadow_rays", "paths"):
                        assert st[k] == out[mode][1][k], (k, conf, mode)
                else:
                    out[mode] = (img, st)


def test_fast_estimator_f32_and_cuts(root):
    s, ref = load(root, "config/cbox_glossy.conf", width=64, height=64, spp=16)
    c = make_ctx(s)
    a = oracle.Oracle(ref, seed=13, estimator="fast").render()
    b = oracle.Oracle(ref, seed=14, estimator="fast").render()
    tol = max(2 * oracle.rmse_rel(b, a), 1e-3)
    f32, _ = c.render(mode="f32", estimator="fast")
    assert oracle.rmse_rel(f32, a) <= tol
    # batches, sample ranges and row bands cut this estimator's frame exactly as the reference's
    full, _ = c.render(mode="f64", estimator="fast")
    small, st = c.render(mode="f64", estimator="fast", batch_paths=64 * 7)
    assert st["passes"] > 10 and np.array_equal(full, small)
    acc, _ = c.render(mode="f64", estimator="fast", spp=8, sample_offset=0, spp_total=16)
    acc, _ = c.render(mode="f64", estimator="fast", spp=8, sample_offset=8, spp_total=16, accumulate=True, out=acc)
    assert np.array_equal(acc, full)
    parts = [c.render(mode="f64", estimator="fast", row_stride=2, row_phase=k, band=8)[0] for k in range(2)]
    assert np.array_equal(parts[0] + parts[1], full)
    with pytest.raises(pt.PtxError):
        c.render(mode="f64", estimator=1 + 1 if False else "fast", depth=70000)
    # the same cuts on a BVH scene (wavefront tracer: test queue, any-hit launches, gather)
    s, ref = load(root, "config/mis.conf", width=64, height=48, spp=8)
    c = make_ctx(s)
    full, st0 = c.render(mode="f64", estimator="fast")
    assert "k_wf_" in c.last_kernel()
    small, st = c.render(mode="f64", estimator="fast", batch_paths=700)
    assert st["passes"] > 10 and np.array_equal(full, small)
    assert (st["segments"], st["shadow_rays"], st["records"]) == (st0["segments"], st0["shadow_rays"], st0["records"])
    acc, _ = c.render(mode="f64", estimator="fast", spp=4, sample_offset=0, spp_total=8)
    acc, _ = c.render(mode="f64", estimator="fast", spp=4, sample_offset=4, spp_total=8, accumulate=True, out=acc)
    assert np.array_equal(acc, full)
    parts = [c.render(mode="f64", estimator="fast", row_stride=2, row_phase=k, band=8)[0] for k in range(2)]
    assert np.array_equal(parts[0] + parts[1], full)
