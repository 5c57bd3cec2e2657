"""Host-side mirror of the reference's tracer interface over the C ABI.

`PathTracer` keeps the names of the reference class (`cPathTracer(conf)`, `Init`, `Process`,
src/tracing/cPathTracer.hpp:18-21 of the reference) so that code and tests read like the
reference's call sites (src/scene/cRTScene.cpp:27,38,59): construct from a conf path, `Init()`
loads mesh / camera / lights and builds the acceleration structure, `Process()` renders and writes
the PPM named by `PathTracer.ResultPath`.  The operator forms (`RayCast`, `VisTest`,
src/geometry/cAccelStruct.hpp:33-35) are batched over numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import PTX_MODE_F32, PTX_MODE_F64, PtxError, check  # noqa: F401

_MODES = {"f64": PTX_MODE_F64, "f32": PTX_MODE_F32, PTX_MODE_F64: PTX_MODE_F64,
          PTX_MODE_F32: PTX_MODE_F32}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Scene:
    """conf + OBJ/MTL + camera + lights, parsed by the C++ front end."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load_conf(cls, conf_path, root_dir=None, width=0, height=0, spp=0, depth=0, model=None,
                  full_region=False):
        L = _lib.lib()
        ov = _lib.Overrides(int(width or 0), int(height or 0), int(spp or 0), int(depth or 0),
                            model.encode() if model else None, int(bool(full_region)))
        h = C.c_void_p()
        check(L.ptx_scene_load_conf(os.fspath(conf_path).encode(),
                                    os.fspath(root_dir).encode() if root_dir else None,
                                    C.byref(ov), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, verts, normals, mat_ids, mats, lights, view, cam_pos, fov_deg, near,
                    width, height, depth, spp, enable_indirect=True, enable_bary_normal=True):
        L = _lib.lib()
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        n = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 9)
        mi = np.ascontiguousarray(mat_ids, dtype=np.int32)
        m = np.ascontiguousarray(mats, dtype=np.float64).reshape(-1, 12)
        lg = np.ascontiguousarray(lights, dtype=np.float64).reshape(-1, 16)
        vw = np.ascontiguousarray(view, dtype=np.float64).reshape(16)
        cp = np.ascontiguousarray(cam_pos, dtype=np.float64).reshape(-1)[:3].copy()
        h = C.c_void_p()
        check(L.ptx_scene_from_arrays(_ptr(v), _ptr(n), _ptr(mi), v.shape[0], _ptr(m), m.shape[0],
                                      _ptr(lg), lg.shape[0], _ptr(vw), _ptr(cp), float(fov_deg),
                                      float(near), int(width), int(height), int(depth), int(spp),
                                      int(bool(enable_indirect)), int(bool(enable_bary_normal)),
                                      C.byref(h)))
        return cls(h)

    def __del__(self):
        try:
            if self._h:
                _lib.lib().ptx_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def info(self):
        i = _lib.SceneInfo()
        check(_lib.lib().ptx_scene_get_info(self._h, C.byref(i)))
        return i

    def geometry(self):
        i = self.info
        verts = np.zeros((i.n_tris, 3, 3), dtype=np.float32)
        normals = np.zeros((i.n_tris, 3, 3), dtype=np.float32)
        mat_ids = np.zeros(i.n_tris, dtype=np.int32)
        check(_lib.lib().ptx_scene_copy_geometry(self._h, _ptr(verts), _ptr(normals), _ptr(mat_ids)))
        return verts, normals, mat_ids

    def vertex_ids(self):
        ids = np.zeros((self.info.n_tris, 3), dtype=np.int32)
        check(_lib.lib().ptx_scene_copy_vertex_ids(self._h, _ptr(ids)))
        return ids

    def tables(self):
        i = self.info
        mats = np.zeros((i.n_mats, 12))
        lights = np.zeros((i.n_lights, 16))
        check(_lib.lib().ptx_scene_copy_tables(self._h, _ptr(mats), _ptr(lights)))
        return mats, lights

    def camera(self):
        view, pos, r2w = np.zeros((4, 4)), np.zeros(3), np.zeros((4, 4))
        check(_lib.lib().ptx_scene_copy_camera(self._h, _ptr(view), _ptr(pos), _ptr(r2w)))
        return view, pos, r2w

    @property
    def result_path(self):
        return _lib.lib().ptx_scene_result_path(self._h).decode()

    def plane_records(self):
        """Plane records of the all-triangles intersector: (n_records, n_pairs, rec_tri[64], blockers[32])."""
        n = np.zeros(2, dtype=np.int32)
        rec_tri = np.zeros(64, dtype=np.int32)
        blockers = np.zeros(32, dtype=np.uint32)
        check(_lib.lib().ptx_scene_plane_records(self._h, _ptr(n[0:1]), _ptr(n[1:2]), _ptr(rec_tri), _ptr(blockers)))
        return int(n[0]), int(n[1]), rec_tri, blockers

    def bvh_check(self, median=False, split=False):
        """Builds the 4-wide BVH on the host and checks its invariants (no GPU needed).  median: median
        splits instead of SAH; split: SAH with reference splitting (the option "bvh_split" at 30)."""
        info = np.zeros(6, dtype=np.int64)
        check(_lib.lib().ptx_scene_bvh_check(self._h, 1 if median else (2 if split else 0), _ptr(info)))
        return dict(n_nodes=int(info[0]), n_children=int(info[1]), max_depth=int(info[2]), stack_need=int(info[3]),
                    triangle_refs=int(info[4]), violations=int(info[5]))


class Context:
    """One GPU: device copy of a scene plus wavefront state."""

    def __init__(self, device=0, stream=None, brute_force_limit=None, batch_paths=None, chunk_paths=None, trace_blocks_per_cu=None):
        L = _lib.lib()
        h = C.c_void_p()
        check(L.ptx_context_create(int(device), C.byref(h)))
        self._h = h
        self._scene = None
        if stream is not None:
            check(L.ptx_context_set_stream(self._h, C.c_void_p(int(stream))))
        if brute_force_limit is not None:
            check(L.ptx_context_set_option(self._h, b"brute_force_limit", int(brute_force_limit)))
        if batch_paths is not None:
            check(L.ptx_context_set_option(self._h, b"default_batch_paths", int(batch_paths)))
        if trace_blocks_per_cu is not None:
            check(L.ptx_context_set_option(self._h, b"trace_blocks_per_cu", int(trace_blocks_per_cu)))
        if chunk_paths is not None:
            check(L.ptx_context_set_option(self._h, b"chunk_paths", int(chunk_paths)))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().ptx_context_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_stream(self, stream):
        check(_lib.lib().ptx_context_set_stream(self._h, C.c_void_p(int(stream or 0))))

    def build_accel(self, scene: Scene):
        check(_lib.lib().ptx_build_accel(self._h, scene._h))
        self._scene = scene
        i = scene.info
        self.width, self.height = i.width, i.height

    def set_option(self, name, value):
        """Tunables of ptx_context_set_option (set before build_accel)."""
        check(_lib.lib().ptx_context_set_option(self._h, name.encode(), int(value)))

    def accel_info(self):
        a = np.zeros(5, dtype=np.int32)
        check(_lib.lib().ptx_accel_info(self._h, _ptr(a)))
        return dict(n_nodes=int(a[0]), n_leaves=int(a[1]), max_depth=int(a[2]),
                    stack_need=int(a[3]), brute_force=bool(a[4]))

    def generate_rays(self):
        n = self.width * self.height
        o, d = np.zeros((n, 3)), np.zeros((n, 3))
        check(_lib.lib().ptx_generate_rays(self._h, _ptr(o), _ptr(d)))
        return o, d

    def first_hit(self, with_points=False):
        n = self.width * self.height
        ids = np.zeros(n, dtype=np.int32)
        pts = np.zeros((n, 3)) if with_points else None
        check(_lib.lib().ptx_first_hit(self._h, _ptr(ids), _ptr(pts)))
        return (ids, pts) if with_points else ids

    def raycast(self, rays_o, rays_d, with_points=True):
        o = np.ascontiguousarray(rays_o, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(rays_d, dtype=np.float64).reshape(-1, 3)
        ids = np.zeros(o.shape[0], dtype=np.int32)
        pts = np.zeros((o.shape[0], 3)) if with_points else None
        check(_lib.lib().ptx_raycast(self._h, _ptr(o), _ptr(d), o.shape[0], _ptr(ids), _ptr(pts)))
        return (ids, pts) if with_points else ids

    def vistest(self, p1, p2):
        a = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(p2, dtype=np.float64).reshape(-1, 3)
        vis = np.zeros(a.shape[0], dtype=np.int32)
        check(_lib.lib().ptx_vistest(self._h, _ptr(a), _ptr(b), a.shape[0], _ptr(vis)))
        return vis.astype(bool)

    def _opts(self, mode, spp, sample_offset, spp_total, rows, row_stride, row_phase, band, depth,
              seed, batch_paths, accumulate, profile, estimator="reference", enqueue_only=False):
        o = _lib.RenderOpts()
        o.enqueue_only = int(bool(enqueue_only))
        o.mode = _MODES[mode]
        o.spp, o.sample_offset, o.spp_total = int(spp or 0), int(sample_offset), int(spp_total or 0)
        if rows is not None:
            o.row_begin, o.row_end = int(rows[0]), int(rows[1])
        o.row_stride, o.row_phase, o.band = int(row_stride), int(row_phase), int(band)
        o.depth = int(depth or 0)
        o.seed = int(seed)
        o.batch_paths = int(batch_paths or 0)
        o.accumulate = int(bool(accumulate))
        o.profile = int(bool(profile))
        o.estimator = {"reference": 0, "fast": 1, 0: 0, 1: 1}[estimator]
        return o

    def render(self, mode="f64", spp=None, sample_offset=0, spp_total=None, rows=None,
               row_stride=1, row_phase=0, band=1, depth=None, seed=13, batch_paths=None,
               accumulate=False, profile=False, out=None, estimator="reference"):
        """Renders into a host array [H,W,3] float64 of pre-final-clamp sums; returns (sum, stats).
        estimator: "reference" (the reference's own) or "fast" (the physically based one)."""
        if out is None:
            out = np.zeros((self.height, self.width, 3))
        o = self._opts(mode, spp, sample_offset, spp_total, rows, row_stride, row_phase, band,
                       depth, seed, batch_paths, accumulate, profile, estimator)
        st = _lib.Stats()
        check(_lib.lib().ptx_render(self._h, C.byref(o), _ptr(out), 0, C.byref(st)))
        return out, _stats_dict(st)

    def render_device(self, device_ptr, mode="f64", spp=None, sample_offset=0, spp_total=None,
                      rows=None, row_stride=1, row_phase=0, band=1, depth=None, seed=13,
                      batch_paths=None, accumulate=False, profile=False, estimator="reference"):
        """Same, into device memory (H*W*3 doubles) owned by the caller, e.g. a torch tensor's
        data_ptr(); nothing is copied to the host."""
        o = self._opts(mode, spp, sample_offset, spp_total, rows, row_stride, row_phase, band,
                       depth, seed, batch_paths, accumulate, profile, estimator)
        st = _lib.Stats()
        check(_lib.lib().ptx_render(self._h, C.byref(o), C.c_void_p(int(device_ptr)), 1, C.byref(st)))
        return _stats_dict(st)


    def render_static(self, device_ptr, mode="f64", spp=None, sample_offset=0, spp_total=None,
                      rows=None, row_stride=1, row_phase=0, band=1, depth=None, seed=13,
                      batch_paths=None, accumulate=False, estimator="reference", enqueue_only=False):
        """render_device through ptx_render_static: the path state lives in the library's static
        device region and the call allocates nothing.  enqueue_only: it does not wait for the stream
        or read anything back either (no statistics) -- what a stream capture needs."""
        o = self._opts(mode, spp, sample_offset, spp_total, rows, row_stride, row_phase, band,
                       depth, seed, batch_paths, accumulate, False, estimator, enqueue_only)
        st = _lib.Stats()
        check(_lib.lib().ptx_render_static(self._h, C.byref(o), C.c_void_p(int(device_ptr)), C.byref(st)))
        return _stats_dict(st)

    def last_kernel(self):
        return _lib.lib().ptx_context_last_kernel(self._h).decode()


def _stats_dict(st):
    return {f[0]: getattr(st, f[0]) for f in _lib.Stats._fields_}


def render_sharded(contexts, scheme="bands", exchange="rccl", mode="f64", spp=None, spp_total=None, rows=None, band=8,
                   depth=None, seed=13, batch_paths=None):
    """One frame over several contexts (one per GPU, one host thread each inside the library) with
    the final reduce of the radiance sums onto the first context's GPU: ptx_render_sharded.
    scheme "samples" | "bands", exchange "rccl" | "peer".  Returns (sum [H,W,3], [stats per context])."""
    ctxs = list(contexts)
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    o = ctxs[0]._opts(mode, spp, 0, spp_total, rows, 1, 0, band, depth, seed, batch_paths, False, False)
    out = np.zeros((ctxs[0].height, ctxs[0].width, 3))
    st = (_lib.Stats * len(ctxs))()
    st[0].struct_size = C.sizeof(_lib.Stats)      # an array is made without running the elements' constructor
    check(_lib.lib().ptx_render_sharded(arr, len(ctxs), C.byref(o), {"samples": 0, "bands": 1}[scheme],
                                        {"rccl": 0, "peer": 1}[exchange], _ptr(out), st))
    return out, [_stats_dict(x) for x in st]


def finalize_image(img_sum):
    a = np.ascontiguousarray(img_sum, dtype=np.float64)
    out = np.zeros_like(a)
    check(_lib.lib().ptx_finalize_image(_ptr(a), a.size // 3, _ptr(out)))
    return out


def write_ppm(path, rgb):
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w = a.shape[0], a.shape[1]
    check(_lib.lib().ptx_write_ppm(os.fspath(path).encode(), _ptr(a), w, h))


def read_ppm(path):
    """P3 reader (the reference's output format); returns uint8-valued ints [H,W,3]."""
    with open(path, "r") as fh:
        tok = fh.read().split()
    if tok[0] != "P3":
        raise ValueError("not a P3 PPM")
    w, h, mx = int(tok[1]), int(tok[2]), int(tok[3])
    data = np.array(tok[4:4 + w * h * 3], dtype=np.int64).reshape(h, w, 3)
    if mx != 255:
        data = data * 255 // mx
    return data


class PathTracer:
    """`cPathTracer` look-alike: PathTracer(conf); Init(); Process()."""

    def __init__(self, conf, root_dir=None, device=0, mode="f64", **overrides):
        self.mConfPath = os.fspath(conf)
        self.root_dir = root_dir
        self.device = device
        self.mode = mode
        self.overrides = overrides
        self.scene = None
        self.ctx = None
        self.image = None
        self.stats = None

    def Init(self, stream=None):
        """mesh + camera + lights + BxDF tables + acceleration structure (cPathTracer::Init)."""
        self.scene = Scene.load_conf(self.mConfPath, self.root_dir, **self.overrides)
        self.ctx = Context(self.device, stream=stream)
        self.ctx.build_accel(self.scene)
        return self

    def Process(self, output=True, **render_kw):
        """GenerateRay + RayTracing + OutputImage (cPathTracer::Process)."""
        if self.ctx is None:
            raise PtxError(5, "state: cPathTracer::Process before Init")
        render_kw.setdefault("mode", self.mode)
        img_sum, self.stats = self.ctx.render(**render_kw)
        self.image = finalize_image(img_sum)
        if output:
            path = self.scene.result_path
            if self.root_dir and not os.path.isabs(path):
                path = os.path.join(self.root_dir, path)
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            write_ppm(path, self.image)
            self.mResultPath = path
        return self.image

    # operator forms
    def GenerateRay(self):
        return self.ctx.generate_rays()

    def RayCast(self, rays_o, rays_d):
        return self.ctx.raycast(rays_o, rays_d)

    def VisTest(self, p1, p2):
        return self.ctx.vistest(p1, p2)
