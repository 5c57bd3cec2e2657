"""ptx_render_static -- the entry point that allocates nothing -- through the C ABI, against
ptx_render on the same inputs: bit for bit, on calls in a row that reuse the library's static state
region with different layouts (a large frame on the all-triangles intersector, then a smaller
frame of a BVH scene through the ring tracer, then a few rows of the first again)."""
import os

import numpy as np
import pytest
import torch

import pathtracer_amd as pt
from test_gpu_parity import load, make_ctx

pytestmark = pytest.mark.gpu


def both(c, shape, **kw):
    a = torch.full(shape, -1.0, dtype=torch.float64, device="cuda")
    b = torch.full(shape, -2.0, dtype=torch.float64, device="cuda")
    st_a = c.render_static(a.data_ptr(), **kw)
    st_b = c.render_device(b.data_ptr(), **kw)
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy(), st_a, st_b


def test_static_state_render_equals_allocating_render(root):
    s1, _ = load(root, "config/cbox.conf", width=320, height=256, spp=8)
    s2, _ = load(root, "config/mis.conf", width=96, height=64, spp=3)
    c1, c2 = make_ctx(s1), make_ctx(s2)
    assert c1.accel_info()["brute_force"] and not c2.accel_info()["brute_force"]
    for mode in ("f64", "f32"):
        a, b, sa, sb = both(c1, (256, 320, 3), mode=mode)                      # first call: the larger input
        assert np.array_equal(a, b) and a.max() > 0
        assert all(sa[k] == sb[k] for k in ("paths", "segments", "shadow_rays", "records"))
        assert 0 < sa["state_bytes"] <= 256 << 20
        a, b, sa, sb = both(c2, (64, 96, 3), mode=mode)                        # second call: a smaller one, another layout
        assert np.array_equal(a, b) and a.max() > 0
        assert all(sa[k] == sb[k] for k in ("paths", "segments", "shadow_rays", "records", "bvh_nodes", "bvh_tris"))
        a, b, sa, sb = both(c1, (256, 320, 3), mode=mode, rows=(100, 108), spp=2)
        assert np.array_equal(a, b) and sa["paths"] == 8 * 320 * 2
    # nothing is allocated: the device's free memory is the same before and after a call
    out = torch.zeros((256, 320, 3), dtype=torch.float64, device="cuda")
    c1.render_static(out.data_ptr(), mode="f64")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    c1.render_static(out.data_ptr(), mode="f64", spp=3)
    c2_out = torch.zeros((64, 96, 3), dtype=torch.float64, device="cuda")
    free0 = torch.cuda.mem_get_info()[0]
    c2.render_static(c2_out.data_ptr(), mode="f64")
    c1.render_static(out.data_ptr(), mode="f32")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    # accumulate continues the sums already in the buffer, as ptx_render does
    full, _ = c1.render(mode="f64")
    c1.render_static(out.data_ptr(), mode="f64", spp=4, sample_offset=0, spp_total=8)
    c1.render_static(out.data_ptr(), mode="f64", spp=4, sample_offset=4, spp_total=8, accumulate=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full)


def test_static_state_render_with_visibility_tests_in_the_rings(root):
    """The physically based estimator on a BVH scene (every light tested: the longer ray ring) and a
    BVH scene with square lights under the reference's estimator: the static region holds a smaller
    grid of them, with the same results as the allocating entry point."""
    s, _ = load(root, "config/mis.conf", width=96, height=64, spp=3)
    for wavefront in (0, 1):                                                   # ring tracer; wavefront tracer (test queue, any-hit launches)
        c = pt.Context(0)
        c.set_option("bvh_wavefront", wavefront)
        c.build_accel(s)
        for mode in ("f64", "f32"):
            a, b, sa, sb = both(c, (64, 96, 3), mode=mode, estimator="fast")
            assert c.last_kernel().startswith("k_wf_traverse" if wavefront else "k_trace_rings_fast")
            assert sa["state_bytes"] <= 256 << 20
            if mode == "f64":
                assert np.abs(a - b).max() <= 1e-9 * max(1.0, float(b.max())) and a.max() > 0
            assert all(sa[k] == sb[k] for k in ("paths", "segments", "shadow_rays", "records"))
    s2, _ = load(root, "config/cbox.conf", width=64, height=48, spp=4)
    for wavefront in (0, 1):
        c2 = pt.Context(0, brute_force_limit=0)                                # the Cornell box on the BVH: its square light is tested through the rings / the test queue
        c2.set_option("bvh_wavefront", wavefront)
        c2.build_accel(s2)
        a, b, sa, sb = both(c2, (48, 64, 3), mode="f64")
        assert c2.last_kernel() == ("k_wf_traverse_f64<0> + k_wf_anyhit" if wavefront else "k_trace_rings_f64<true, true>")
        assert np.array_equal(a, b) and sa["shadow_rays"] == sb["shadow_rays"] > 0


def test_static_state_render_rejects_what_does_not_fit(root):
    s, _ = load(root, "config/cbox.conf", width=1024, height=64, spp=1)
    c = make_ctx(s)
    out = torch.zeros((64, 1024, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(pt.PtxError) as e:
        c.render_static(out.data_ptr(), mode="f64", depth=60000)               # 192 slots x 60000 bounces x 64 B per wave
    assert e.value.code == 1 and "static state region" in e.value.message
    st = c.render_static(out.data_ptr(), mode="f64", depth=200)                # a deep path still fits on a smaller grid
    ref = torch.zeros_like(out)
    st2 = c.render_device(ref.data_ptr(), mode="f64", depth=200)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and st["segments"] == st2["segments"]
    with pytest.raises(pt.PtxError):
        _lib_null = pt.lib().ptx_render_static(c._h, None, None, None)
        pt._lib.check(_lib_null)


@pytest.mark.parametrize("conf,kw,depth", [("config/cbox.conf", dict(width=96, height=80, spp=4), None),
                                           ("config/mis.conf", dict(width=96, height=64, spp=3), None),
                                           ("config/mis.conf", dict(width=48, height=32, spp=2), 20)])
def test_static_render_under_stream_capture(root, conf, kw, depth):
    """opts.enqueue_only: ptx_render_static neither waits for its stream nor reads anything back, so a
    stream capture takes it -- the all-triangles tracer, the wavefront tracer of a BVH scene, and that
    tracer at a depth where it would otherwise read a queue count back every eighth bounce.  The
    captured graph is launched twice; both frames equal ptx_render's bit for bit."""
    s, _ = load(root, conf, **kw)
    c = make_ctx(s)
    H, W = kw["height"], kw["width"]
    want, _ = c.render(mode="f64", depth=depth)
    out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    c.render_static(out.data_ptr(), mode="f64", depth=depth)               # kernels loaded, LDS limits set: not part of the capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    c.set_stream(side.cuda_stream)
    try:
        with torch.cuda.graph(g, stream=side):
            st = c.render_static(out.data_ptr(), mode="f64", depth=depth, enqueue_only=True)
    finally:
        c.set_stream(None)
    assert st["paths"] == 0 and st["passes"] >= 1 and st["state_bytes"] > 0   # no statistics came back: nothing was waited for
    for _ in range(2):
        out.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    # outside a capture the flag only drops the wait: the frame is there once the stream is done
    out.fill_(-1.0)
    c.render_static(out.data_ptr(), mode="f64", depth=depth, enqueue_only=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    import ctypes
    host = np.zeros((H, W, 3))
    o = c._opts("f64", None, 0, None, None, 1, 0, 1, None, 13, None, False, False, "reference", True)
    rc = pt.lib().ptx_render(c._h, ctypes.byref(o), ctypes.c_void_p(host.ctypes.data), 0, None)
    assert rc == 1 and b"enqueue_only" in pt.lib().ptx_last_error()       # a host buffer would have to be waited for
