"""Image tooling (PPM statistics / compare / merge) and the PPM writer's byte format, pinned by the
one raw output the reference holds: results/mis.ppm (768 x 512, written by
cPathTracer::OutputImage, src/tracing/cPathTracer.cpp:261-284 of the reference)."""
import hashlib
import json
import os

import numpy as np
import pytest

import pathtracer_amd as pt
from pathtracer_amd import image

REF_PPM = "/root/reference/results/mis.ppm"


@pytest.fixture(scope="module")
def fix(root):
    return json.load(open(os.path.join(root, "tests/golden/mis_ppm_stats.json")))


def test_writer_reproduces_the_reference_bytes_of_the_fixture_pixels(fix, tmp_path):
    px = np.array(fix["first_pixels"], dtype=np.float64).reshape(1, -1, 3)
    p = tmp_path / "row.ppm"
    pt.write_ppm(str(p), (px + 0.5) / 255.0)
    text = p.read_text()
    assert text == "P3\n%d 1\n255\n" % px.shape[1] + fix["first_pixels_text"]
    assert fix["header"] == "P3\n768 512\n255\n" and fix["newlines"] == 3 and fix["ends_with"].endswith(" ")
    assert np.array_equal(pt.read_ppm(str(p)), px.astype(np.int64))


@pytest.mark.skipif(not os.path.exists(REF_PPM), reason="reference checkout not mounted")
def test_reference_image_round_trips_byte_for_byte(fix, tmp_path):
    raw = open(REF_PPM, "rb").read()
    assert hashlib.sha256(raw).hexdigest() == fix["sha256"] and len(raw) == fix["bytes"]
    img = pt.read_ppm(REF_PPM)
    st = image.image_stats(img)
    assert st == fix["stats"] and (st["height"], st["width"]) == (512, 768)
    p = tmp_path / "again.ppm"
    pt.write_ppm(str(p), image.to_linear(img))
    assert open(p, "rb").read() == raw


def test_compare_and_stats(tmp_path):
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 1, (6, 5, 3))
    b = a.copy()
    assert image.compare(a, b)["relative_mse"] == 0.0 and image.compare(a, b)["psnr_db"] == float("inf")
    b[2, 3, 1] += 0.5
    c = image.compare(b, a)
    want = 0.25 / (a[2, 3, 1] ** 2 + 1e-2) / 30
    assert abs(c["relative_mse"] - want) < 1e-15 and c["max_at_row_col"] == [2, 3] and c["differing_pixels"] == 1
    with pytest.raises(ValueError):
        image.relative_mse(a, a[:3])
    pa, pb = tmp_path / "a.ppm", tmp_path / "b.ppm"
    pt.write_ppm(str(pa), a), pt.write_ppm(str(pb), b)
    st = image.image_stats(pt.read_ppm(str(pa)))
    assert st == image.image_stats(a)                       # float images are quantised like the writer does
    assert image.main([str(pa), str(pa), "--tolerance", "1e-9"]) == 0
    assert image.main([str(pb), str(pa), "--tolerance", "1e-9"]) == 1
    assert image.main([]) == 2


def test_merge_of_partial_sums_is_the_whole_frame():
    rng = np.random.default_rng(5)
    full = rng.uniform(0, 3, (16, 4, 3))
    bands = []
    for k in range(2):                                      # interleaved bands of 4 rows
        part = np.zeros_like(full)
        for r in range(16):
            if (r // 4) % 2 == k:
                part[r] = full[r]
        bands.append(part)
    assert np.array_equal(image.merge_partial_sums(bands), full)
    halves = [full * 0.25, full * 0.75]                     # sample ranges at the same spp_total
    assert np.allclose(image.merge_partial_sums(halves), full, rtol=0, atol=1e-15)
    assert np.array_equal(pt.finalize_image(image.merge_partial_sums(bands)), pt.finalize_image(full))


def test_colour_difference_and_perceptual_map():
    from pathtracer_amd import image
    rng = np.random.default_rng(0)
    a = rng.uniform(0, 1, (24, 32, 3))
    assert np.allclose(image.to_lab(np.ones((1, 1, 3))), [[[100.0, 0.0, 0.0]]], atol=1e-3)      # white point
    assert np.allclose(image.to_lab(np.zeros((1, 1, 3))), 0.0, atol=1e-9)
    assert image.delta_e(a, a).max() == 0 and image.flip_lite(a, a).max() == 0
    # a uniform luminance step of 1 % is far below, inverting the image far above a just noticeable difference
    assert image.delta_e(a * 0.99, a).mean() < 1.0 < 2.3 < image.delta_e(1 - a, a).mean()
    noisy = np.clip(a + rng.normal(0, 0.05, a.shape), 0, 1)
    shifted = np.clip(a + 0.05, 0, 1)
    # the blur makes zero-mean noise weigh less than a shift of the same size
    assert image.flip_lite(noisy, a).mean() < image.flip_lite(shifted, a).mean()
    c = image.compare((a * 255).astype(np.int64), (shifted * 255).astype(np.int64))
    assert {"delta_e76_mean", "delta_e76_p95", "delta_e76_max", "flip_lite_mean"} <= set(c) and c["delta_e76_mean"] > 0


def test_merge_of_part_files(tmp_path):
    """Part files (the pre-clamp sums of some rows x some samples of a frame, `pathtracer --dump-partial`)
    merge to the frame only when they tile it: overlaps, holes and parts of another frame are refused."""
    from pathtracer_amd import image
    rng = np.random.default_rng(5)
    H, W, S = 12, 7, 8
    per_sample = rng.uniform(0, 2, (S, H, W, 3)) / S          # what each sample adds to the sums
    full = per_sample.sum(0)

    def part(name, rows, samples, **kw):
        p = str(tmp_path / name)
        image.write_part(p, per_sample[samples[0]:samples[1], rows[0]:rows[1]].sum(0), W, H, S, rows, samples, depth=5, **kw)
        return p

    a, b = part("top", (0, 5), (0, S)), part("bottom", (5, H), (0, S))
    total, head = image.merge_parts([b, a])
    assert np.allclose(total, full, rtol=0, atol=1e-15) and (head["width"], head["height"], head["spp_total"]) == (W, H, S)
    quads = [part("q%d" % i, r, s) for i, (r, s) in enumerate([((0, 5), (0, 3)), ((0, 5), (3, S)), ((5, H), (0, 6)), ((5, H), (6, S))])]
    assert np.allclose(image.merge_parts(quads)[0], full, rtol=0, atol=1e-15)
    h, rows_sum = image.read_part(a)
    assert h["rows"] == (0, 5) and h["samples"] == (0, S) and h["seed"] == 13 and rows_sum.shape == (5, W, 3)
    for bad, why in (([a], "cover"), ([a, b, quads[0]], "overlaps"), ([a, part("other_seed", (5, H), (0, S), seed=14)], "seed"),
                     ([a, part("f32", (5, H), (0, S), mode=1)], "mode"), ([quads[0], quads[2], quads[3]], "cover")):
        with pytest.raises(ValueError) as e:
            image.merge_parts(bad)
        assert why in str(e.value)
    (tmp_path / "junk").write_bytes(b"P3\n1 1\n255\n0 0 0 ")
    with pytest.raises(ValueError):
        image.read_part(str(tmp_path / "junk"))
    open(a, "ab").write(b"\0" * 8)                              # a truncated or padded file is not read
    with pytest.raises(ValueError):
        image.read_part(a)
    # the command line: exit code 1 and the reference's "[error]" convention on a refusal, 2 on misuse
    assert image.main(["--merge", b, "-o", str(tmp_path / "x.ppm")]) == 1
    assert image.main(["--merge", b]) == 2
