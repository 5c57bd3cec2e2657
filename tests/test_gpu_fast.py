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
    ("config/mis.conf", 96, 64, 4, {}),                        # five sphere lights, each with its visibility test
    ("config/room.conf", 96, 64, 2, {})])
def test_fast_estimator_f64_matches_oracle(root, conf, w, h, spp, kw):
    s, ref = load(root, conf, width=w, height=h, spp=spp)
    c = pt.Context(0, brute_force_limit=kw.get("brute_force_limit"))
    if "shadow_queue" in kw:
        c.set_option("shadow_queue", kw["shadow_queue"])
    c.build_accel(s)
    O = oracle.Oracle(ref, seed=13, estimator="fast")
    img, st = c.render(mode="f64", seed=13, estimator="fast")
    oimg, ost = O.render(stats=True)
    assert c.last_kernel().startswith("k_trace_fast_f64")
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
