"""ptx_render_sharded and the command line's --devices: one frame over several contexts driven by
host threads inside the library, and the final exchange.  A one-GPU box can run several contexts on
its one GPU with the peer-copy exchange (same threads, same cuts, same ordered sum) and RCCL with a
communicator of one; RCCL between GPUs needs a node with more than one."""
import os
import subprocess

import numpy as np
import pytest

import pathtracer_amd as pt
from test_gpu_parity import load, make_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conf,kw", [("config/cbox.conf", dict(width=96, height=80, spp=7)),
                                     ("config/mis.conf", dict(width=96, height=64, spp=5))])
def test_sharded_render_equals_single_render(root, conf, kw):
    s, _ = load(root, conf, **kw)
    ctxs = [make_ctx(s) for _ in range(3)]
    full, st = ctxs[0].render(mode="f64")
    bands, sb = pt.render_sharded(ctxs, scheme="bands", exchange="peer", band=8)
    assert np.array_equal(bands, full)                               # disjoint rows: the frame, bit for bit
    assert sum(x["paths"] for x in sb) == st["paths"] and sum(x["segments"] for x in sb) == st["segments"]
    assert all(x["paths"] > 0 for x in sb)
    samples, ss = pt.render_sharded(ctxs, scheme="samples", exchange="peer")
    assert [x["paths"] for x in ss] == [kw["width"] * kw["height"] * n for n in
                                        (kw["spp"] // 3 + (kw["spp"] % 3 > 0), kw["spp"] // 3 + (kw["spp"] % 3 > 1), kw["spp"] // 3)]
    assert np.abs(samples - full).max() <= 1e-14 * max(1.0, full.max())
    # more contexts than samples: the idle ones contribute zeros
    two, s2 = pt.render_sharded(ctxs, scheme="samples", exchange="peer", spp=2)
    ref2, _ = ctxs[0].render(mode="f64", spp=2)
    assert s2[2]["paths"] == 0 and np.abs(two - ref2).max() <= 1e-14 * max(1.0, ref2.max())
    f32, _ = pt.render_sharded(ctxs[:2], scheme="bands", exchange="peer", mode="f32")
    assert np.array_equal(f32, ctxs[0].render(mode="f32")[0])


def test_band_gather_ragged_frames_and_reuse(root):
    """Row bands: only a context's own rows travel (packed, one message per context) and the first
    context puts them in place -- whatever the height (not a multiple of band x contexts, fewer bands
    than contexts), the band height and the number of contexts, and frame after frame on the same
    contexts (their buffers and host threads are kept): always the single-context frame bit for bit."""
    import json
    base = json.load(open(os.path.join(root, "config/cbox.conf")))
    for w, h in ((40, 61), (24, 13), (56, 96), (16, 7)):          # the larger frame after smaller ones and back
        s, _ = load(root, "config/cbox.conf", width=w, height=h, spp=3)
        ctxs = [make_ctx(s) for _ in range(4)]
        full, st = ctxs[0].render(mode="f64")
        for n, band in ((4, 8), (3, 5), (2, 1), (4, 3)):
            for _ in range(2):                                     # twice: the second frame reuses everything
                img, sb = pt.render_sharded(ctxs[:n], scheme="bands", exchange="peer", band=band)
                assert np.array_equal(img, full), (w, h, n, band)
                assert sum(x["paths"] for x in sb) == st["paths"]
        # a caller's own interleave options do not leak into the sharded call
        o_img, _ = pt.render_sharded(ctxs[:2], scheme="samples", exchange="peer")
        assert np.abs(o_img - full).max() <= 1e-14 * max(1.0, full.max())


def test_rccl_exchange_with_a_communicator_of_one(root):
    s, _ = load(root, "config/cbox.conf", width=64, height=64, spp=4)
    c = make_ctx(s)
    full, _ = c.render(mode="f64")
    for scheme in ("bands", "samples"):
        img, st = pt.render_sharded([c], scheme=scheme, exchange="rccl")
        assert np.array_equal(img, full)
    with pytest.raises(pt.PtxError) as e:
        pt.render_sharded([c, make_ctx(s)], exchange="rccl")          # two contexts on one GPU: not a communicator
    assert e.value.code == 1


def test_command_line_shards_over_devices(root, tmp_path):
    exe = os.path.join(root, "pathtracer_amd", "pathtracer")
    base = [exe, "config/cbox.conf", "--width", "72", "--height", "56", "--spp", "5"]
    outs = {}
    for name, extra in (("one", []), ("bands", ["--devices", "0,0,0", "--exchange", "peer"]),
                        ("samples", ["--devices", "0,0", "--exchange", "peer", "--scheme", "samples"]),
                        ("rccl1", ["--gpus", "1"])):
        out = tmp_path / (name + ".ppm")
        p = subprocess.run(base + extra + ["--out", str(out)], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stdout
        outs[name] = pt.read_ppm(str(out))
    assert np.array_equal(outs["one"], outs["bands"]) and np.array_equal(outs["one"], outs["rccl1"])
    assert np.abs(outs["one"] - outs["samples"]).max() <= 1            # int(c * 255) of sums that differ by an ulp
    p = subprocess.run(base + ["--devices", "0,7"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 1 and "[error]" in p.stdout                  # the reference's error convention


def test_dumped_parts_merge_to_the_one_piece_ppm(root, tmp_path):
    """`pathtracer <conf> --dump-partial FILE --rows A:B | --samples A:B` writes the pre-clamp sums of a
    part of the frame; `python -m pathtracer_amd.image --merge` adds the parts, clamps and writes the
    PPM.  Two row halves (and four quarters: rows x samples cut at the batch boundaries a single render
    has) give the one-piece file byte for byte; a missing part is refused."""
    import sys
    from pathtracer_amd import image
    exe = os.path.join(root, "pathtracer_amd", "pathtracer")
    base = [exe, "config/mis.conf", "--width", "72", "--height", "56", "--spp", "6"]

    def run(extra):
        p = subprocess.run(base + extra, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert p.returncode == 0, p.stdout
        return p.stdout

    whole = tmp_path / "whole.ppm"
    run(["--out", str(whole)])
    parts = {}
    for name, extra in (("top", ["--rows", "0:24"]), ("bottom", ["--rows", "24:56"]),
                        ("q0", ["--rows", "0:24", "--samples", "0:4"]), ("q1", ["--rows", "0:24", "--samples", "4:6"]),
                        ("q2", ["--rows", "24:56", "--samples", "0:1"]), ("q3", ["--rows", "24:56", "--samples", "1:6"])):
        parts[name] = str(tmp_path / (name + ".part"))
        assert "rows" in run(["--dump-partial", parts[name]] + extra)
    merge = [sys.executable, "-m", "pathtracer_amd.image", "--merge"]
    out = tmp_path / "halves.ppm"
    p = subprocess.run(merge + [parts["top"], parts["bottom"], "-o", str(out)], cwd=root, stdout=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stdout
    assert open(out, "rb").read() == open(whole, "rb").read()                 # byte for byte
    out4 = tmp_path / "quarters.ppm"
    p = subprocess.run(merge + [parts[k] for k in ("q3", "q0", "q2", "q1")] + ["-o", str(out4)], cwd=root, stdout=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 0, p.stdout
    a, b = pt.read_ppm(str(out4)), pt.read_ppm(str(whole))
    assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3               # sample ranges added apart: sums differ by an ulp at most
    total, head = image.merge_parts([parts["top"], parts["bottom"]])
    s, _ = load(root, "config/mis.conf", width=72, height=56, spp=6)
    assert np.array_equal(total, make_ctx(s).render(mode="f64")[0]) and head["depth"] == 5
    p = subprocess.run(merge + [parts["top"], parts["q2"], "-o", str(out)], cwd=root, stdout=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 1 and "[error]" in p.stdout
    p = subprocess.run(base + ["--rows", "0:10"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 1 and "--dump-partial" in p.stdout
    # --resume: the same command line again leaves a complete part alone, renders a missing, a truncated or a foreign one
    before = os.stat(parts["top"]).st_mtime_ns
    assert "nothing to do" in run(["--dump-partial", parts["top"], "--rows", "0:24", "--resume"])
    assert os.stat(parts["top"]).st_mtime_ns == before
    blob = open(parts["top"], "rb").read()
    open(parts["top"], "wb").write(blob[:-8])                                    # cut short
    assert "nothing to do" not in run(["--dump-partial", parts["top"], "--rows", "0:24", "--resume"])
    assert open(parts["top"], "rb").read() == blob                               # and whole again, the same bytes
    assert "nothing to do" not in run(["--dump-partial", parts["top"], "--rows", "0:20", "--resume"])   # another piece: rendered
    assert "nothing to do" not in run(["--dump-partial", str(tmp_path / "new.part"), "--rows", "0:24", "--resume"])
    p = subprocess.run(base + ["--resume"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 1 and "--resume" in p.stdout
