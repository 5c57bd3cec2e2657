"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (pathtracer_amd) never does.  The shared object is built by oracle/Makefile
(``make -C oracle``) from oracle/pt_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OrcScene(C.Structure):
    _fields_ = [
        ("n_tris", C.c_int32), ("verts", C.c_void_p), ("normals", C.c_void_p),
        ("mat_ids", C.c_void_p), ("n_mats", C.c_int32), ("mats", C.c_void_p),
        ("n_lights", C.c_int32), ("lights", C.c_void_p),
        ("width", C.c_int32), ("height", C.c_int32),
        ("fov_deg", C.c_double), ("near_", C.c_double),
        ("view", C.c_double * 16), ("cam_pos", C.c_double * 3),
        ("depth", C.c_int32), ("spp", C.c_int32),
        ("enable_indirect", C.c_int32), ("enable_bary_normal", C.c_int32),
        ("region", C.c_int32 * 4), ("seed", C.c_uint64),
        ("sample_offset", C.c_int32), ("spp_total", C.c_int32), ("use_bvh", C.c_int32),
        ("estimator", C.c_int32), ("fast_strategy", C.c_int32),
    ]


class OrcStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("closest_rays", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("tri_tests", C.c_uint64)]


_lib = None
_variant = 0


def use_sum_order(variant):
    """Switches every later oracle call of this process to the restatement built with ORC_SUM =
    variant (0: left-to-right sums, the default; 1, 2: the other plausible Eigen orders)."""
    global _lib, _variant
    if variant != _variant:
        _lib, _variant = None, int(variant)


def lib():
    global _lib
    if _lib is None:
        path = _LIB_PATH if _variant == 0 else os.path.join(_HERE, "liboracle_sum%d.so" % _variant)
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run `make -C oracle`" % path)
        L = C.CDLL(path)
        assert L.orc_sum_order() == _variant
        L.orc_rng.restype = C.c_double
        L.orc_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_bits.restype = C.c_uint64
        L.orc_rng_bits.argtypes = L.orc_rng.argtypes
        L.orc_tri_raycast.restype = C.c_int
        L.orc_tri_raycast_t.restype = C.c_double
        L.orc_closest_hit.restype = C.c_int32
        L.orc_closest_hit_probe.restype = C.c_int32
        L.orc_trace_sample.restype = C.c_int32
        L.orc_trace_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_sample_ties.restype = C.c_int32
        L.orc_trace_sample_ties.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_double,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_sample_perturbed.restype = C.c_int32
        L.orc_trace_sample_perturbed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        L.orc_vistest.restype = C.c_int
        L.orc_square_light_sample.restype = C.c_double
        L.orc_square_light_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p]
        L.orc_sphere_light_sample.restype = C.c_double
        L.orc_sphere_light_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                              C.c_double, C.c_void_p, C.c_void_p]
        L.orc_sample_hemisphere.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.orc_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_is_bsdf.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_finalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _v3(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1)[:3])


class Oracle:
    """Holds a scene in the layout pt_oracle.c expects and exposes its entry points."""

    def __init__(self, scene, seed=13, use_bvh=None, estimator="reference", fast_strategy=0):
        """estimator: "reference" (the reference's own, what parity is defined against) or "fast"
        (the physically based one: MIS, lobe sampling, Russian roulette, shadowed sphere lights)."""
        self.scene = scene
        self._keep = dict(
            verts=np.ascontiguousarray(scene.verts, dtype=np.float64),
            normals=np.ascontiguousarray(scene.normals, dtype=np.float64),
            mat_ids=np.ascontiguousarray(scene.mat_ids, dtype=np.int32),
            mats=np.ascontiguousarray(scene.mats, dtype=np.float64),
            lights=np.ascontiguousarray(scene.lights, dtype=np.float64),
        )
        k = self._keep
        s = OrcScene()
        s.n_tris = k["verts"].shape[0]
        s.verts, s.normals, s.mat_ids = _p(k["verts"]), _p(k["normals"]), _p(k["mat_ids"])
        s.n_mats, s.mats = k["mats"].shape[0], _p(k["mats"])
        s.n_lights, s.lights = k["lights"].shape[0], _p(k["lights"])
        s.width, s.height = scene.width, scene.height
        s.fov_deg, s.near_ = scene.fov_deg, scene.near
        s.view = (C.c_double * 16)(*np.asarray(scene.view, dtype=np.float64).reshape(-1))
        s.cam_pos = (C.c_double * 3)(*np.asarray(scene.cam_pos, dtype=np.float64).reshape(-1)[:3])
        s.depth, s.spp = scene.depth, scene.spp
        s.enable_indirect = int(scene.enable_indirect)
        s.enable_bary_normal = int(scene.enable_bary_normal)
        s.region = (C.c_int32 * 4)(*scene.region)
        s.seed = seed
        s.sample_offset = 0
        s.spp_total = scene.spp
        if use_bvh is None:
            use_bvh = s.n_tris > 256
        s.use_bvh = int(bool(use_bvh))
        s.estimator = {"reference": 0, "fast": 1}[estimator]
        s.fast_strategy = int(fast_strategy)
        self.c = s
        if s.use_bvh:
            lib().orc_bvh_build(C.byref(s))

    # -- geometry ------------------------------------------------------------------------------
    def generate_rays(self):
        n = self.c.width * self.c.height
        o = np.zeros((n, 3))
        d = np.zeros((n, 3))
        lib().orc_generate_rays(C.byref(self.c), _p(o), _p(d))
        return o, d

    def camera_matrix(self):
        m = np.zeros(16)
        lib().orc_camera_matrix(C.byref(self.c), _p(m))
        return m.reshape(4, 4)

    def first_hit(self, with_points=False):
        n = self.c.width * self.c.height
        ids = np.zeros(n, dtype=np.int32)
        pts = np.zeros((n, 3)) if with_points else None
        lib().orc_first_hit(C.byref(self.c), _p(ids), _p(pts) if with_points else None)
        return (ids, pts) if with_points else ids

    def closest_hit(self, o, d):
        o, d = _v3(o), _v3(d)
        pt = np.zeros(3)
        f = lib().orc_closest_hit(C.byref(self.c), _p(o), _p(d), _p(pt))
        return int(f), pt

    def vistest(self, p1, p2):
        p1, p2 = _v3(p1), _v3(p2)
        return bool(lib().orc_vistest(C.byref(self.c), _p(p1), _p(p2)))

    def closest_hit_probe(self, o, d):
        """closest_hit plus the runner-up's distance (inf: none): (face, point, second)."""
        o, d = _v3(o), _v3(d)
        pt = np.zeros(3)
        second = C.c_double(0.0)
        f = lib().orc_closest_hit_probe(C.byref(self.c), _p(o), _p(d), _p(pt), C.byref(second))
        return int(f), pt, float(second.value)

    def trace_sample(self, x, y, sample):
        """One sample of pixel (column x, row y): (rgb before the division by the sample count,
        faces hit per bounce, margin by which each won its closest-hit search)."""
        rgb = np.zeros(3)
        faces = np.full(64, -2, dtype=np.int32)
        margins = np.full(64, np.inf)
        n = lib().orc_trace_sample(C.byref(self.c), int(x), int(y), int(sample), _p(rgb), _p(faces), _p(margins))
        return rgb, faces[:n].copy(), margins[:n].copy()

    def trace_sample_ties(self, x, y, sample, mask=0, gap=1e-10):
        """trace_sample with the path's ties falling as `mask` says: a closest-hit search whose
        winner and runner-up lie within `gap` is a tie, ties are numbered in the order the path
        meets them, bit k set hands tie k to the runner-up.  Returns (rgb, faces, margins, ties met)."""
        rgb = np.zeros(3)
        faces = np.full(64, -2, dtype=np.int32)
        margins = np.full(64, np.inf)
        nt = C.c_int32(0)
        n = lib().orc_trace_sample_ties(C.byref(self.c), int(x), int(y), int(sample), int(mask), float(gap),
                                        _p(rgb), _p(faces), _p(margins), C.byref(nt))
        return rgb, faces[:n].copy(), margins[:n].copy(), int(nt.value)

    def amplification(self, x, y, sample, eps=1e-9):
        """How strongly the sample's value reacts to an error of one of its own sampled directions:
        max over the bounces b of |value(direction of bounce b turned by eps) - value| / eps (reference's
        estimator only; None otherwise).  1..10 on flat, diffuse geometry; curved, smoothly shaded
        surfaces multiply a direction error at every further bounce."""
        base = np.zeros(3)
        n = lib().orc_trace_sample_perturbed(C.byref(self.c), int(x), int(y), int(sample), -1, 0.0, _p(base))
        if n < 0:
            return None
        worst = 0.0
        for b in range(max(n, 1)):
            rgb = np.zeros(3)
            lib().orc_trace_sample_perturbed(C.byref(self.c), int(x), int(y), int(sample), b, float(eps), _p(rgb))
            worst = max(worst, float(np.abs(rgb - base).max()) / eps)
        return worst

    def tie_alternatives(self, x, y, sample, gap=1e-10, max_ties=6):
        """Every value the sample can take when the ties on its path fall either way: a list of
        (mask, rgb, faces).  Mask 0 is the reference's own rule (lowest face id).  A path that meets
        more than max_ties ties under some mask raises: that is no longer a tie, it is a scene made
        of them."""
        out, seen = [], set()
        pending = [0]
        while pending:
            mask = pending.pop()
            if mask in seen:
                continue
            seen.add(mask)
            rgb, faces, margins, nt = self.trace_sample_ties(x, y, sample, mask, gap)
            if nt > max_ties:
                raise AssertionError("pixel (%d, %d) sample %d meets %d ties on one path (mask %#x)" % (y, x, sample, nt, mask))
            out.append((mask, rgb, faces))
            # a set bit beyond the ties this path met changes nothing; flipping any tie it met is a new path
            for k in range(nt):
                m = (mask & ((1 << (k + 1)) - 1)) ^ (1 << k)
                # ties after k belong to the path as it ran under `mask`; under m they are met afresh
                if m not in seen:
                    pending.append(m)
        return out

    # -- render --------------------------------------------------------------------------------
    def render(self, spp=None, sample_offset=0, spp_total=None, threads=0, stats=False):
        """Returns the pre-final-clamp sums [H,W,3] (and stats when asked)."""
        c = self.c
        if spp is not None:
            c.spp = int(spp)
        c.sample_offset = int(sample_offset)
        c.spp_total = int(spp_total) if spp_total is not None else c.spp
        out = np.zeros((c.height, c.width, 3))
        st = OrcStats()
        lib().orc_render(C.byref(c), _p(out), C.byref(st) if stats else None, int(threads))
        if stats:
            return out, {f[0]: int(getattr(st, f[0])) for f in OrcStats._fields_}
        return out


def finalize(img_sum):
    a = np.ascontiguousarray(img_sum, dtype=np.float64)
    out = np.zeros_like(a)
    lib().orc_finalize(_p(a), _p(out), a.size // 3)
    return out


# -- free functions with explicit inputs (known-answer tests) ----------------------------------
def rng(seed, pixel, sample, bounce, dim):
    return lib().orc_rng(seed, pixel, sample, bounce, dim)


def tri_raycast(o, d, p1, p2, p3):
    o, d, p1, p2, p3 = map(_v3, (o, d, p1, p2, p3))
    pt = np.zeros(3)
    ok = lib().orc_tri_raycast(_p(o), _p(d), _p(p1), _p(p2), _p(p3), _p(pt))
    return bool(ok), pt


def tri_raycast_t(o, d, p1, p2, p3):
    o, d, p1, p2, p3 = map(_v3, (o, d, p1, p2, p3))
    return lib().orc_tri_raycast_t(_p(o), _p(d), _p(p1), _p(p2), _p(p3))


def reflect(n, i):
    n, i = _v3(n), _v3(i)
    o = np.zeros(3)
    lib().orc_reflect(_p(n), _p(i), _p(o))
    return o


def refract(n, i, ni):
    n, i = _v3(n), _v3(i)
    o = np.zeros(3)
    lib().orc_refract(_p(n), _p(i), float(ni), _p(o))
    return o


def barycentric(p, a, b, c):
    p, a, b, c = map(_v3, (p, a, b, c))
    w = np.zeros(3)
    lib().orc_barycentric(_p(p), _p(a), _p(b), _p(c), _p(w))
    return w


def normal_from_3pts(a, b, c):
    a, b, c = map(_v3, (a, b, c))
    n = np.zeros(3)
    lib().orc_normal_from_3pts(_p(a), _p(b), _p(c), _p(n))
    return n


def sample_hemisphere(n, xi1, xi2):
    n = _v3(n)
    o = np.zeros(3)
    lib().orc_sample_hemisphere(_p(n), float(xi1), float(xi2), _p(o))
    return o


def square_light_sample(light16, ref, xi1, xi2):
    L = np.ascontiguousarray(light16, dtype=np.float64)
    ref = _v3(ref)
    ro, rd = np.zeros(3), np.zeros(3)
    pdf = lib().orc_square_light_sample(_p(L), _p(ref), float(xi1), float(xi2), _p(ro), _p(rd))
    return pdf, ro, rd


def sphere_light_sample(light16, ref, xi1, xi2, xi3):
    L = np.ascontiguousarray(light16, dtype=np.float64)
    ref = _v3(ref)
    ro, rd = np.zeros(3), np.zeros(3)
    pdf = lib().orc_sphere_light_sample(_p(L), _p(ref), float(xi1), float(xi2), float(xi3),
                                        _p(ro), _p(rd))
    return pdf, ro, rd


def bxdf_eval(mat12, wi, wo, n):
    m = np.ascontiguousarray(mat12, dtype=np.float64)
    wi, wo, n = map(_v3, (wi, wo, n))
    f = np.zeros(3)
    lib().orc_bxdf_eval(_p(m), _p(wi), _p(wo), _p(n), _p(f))
    return f


def bxdf_sample(mat12, n, wo, xi):
    m = np.ascontiguousarray(mat12, dtype=np.float64)
    n, wo = _v3(n), _v3(wo)
    x = np.ascontiguousarray(np.asarray(xi, dtype=np.float64).reshape(-1)[:2])
    if x.size < 2:
        x = np.concatenate([x, np.zeros(2 - x.size)])
    wi, coef = np.zeros(3), np.zeros(3)
    lib().orc_bxdf_sample(_p(m), _p(n), _p(wo), _p(x), _p(wi), _p(coef))
    return wi, coef


def rmse_rel(a, b):
    """Per-pixel relative MSE  mean_p sum_c (a-b)^2 / (b^2 + 1e-2)  on linear float images."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.mean(np.sum((a - b) ** 2 / (b ** 2 + 1e-2), axis=-1)))


class RefAccel:
    """One of the reference's two culling structures over a triangle array (oracle/ref_accels.c):
    RefAccel(verts, "AABB", subdivide) is cAABB, RefAccel(verts, "OCTREE", capacity) is cOctree
    (src/geometry/cAccelStruct.cpp:89-263, 266-482, 614-730 of the reference)."""

    def __init__(self, verts, kind, param, octree_file=None):
        L = lib()
        L.orc_octree_write.restype = C.c_int
        L.orc_octree_write.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_octree_read.restype = C.c_void_p
        L.orc_octree_read.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p]
        L.orc_accel_build.restype = C.c_void_p
        L.orc_accel_build.argtypes = [C.c_void_p, C.c_int32, C.c_int, C.c_int]
        L.orc_accel_free.argtypes = [C.c_void_p]
        L.orc_accel_info.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_accel_raycast_many.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_accel_vistest_many.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        self.verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        self.kind = kind
        if octree_file is not None:
            # cOctree::ReadOctree (src/geometry/cAccelStruct.cpp:524-611): the structure as its cache holds it
            st = C.c_int(0)
            self._h = L.orc_octree_read(_p(self.verts), self.verts.shape[0], os.fsencode(octree_file), C.byref(st))
            if not self._h:
                raise RuntimeError({1: "cannot open %s" % octree_file, 2: "node id beyond the reader's buffer (the reference exits here)",
                                    3: "malformed .octree file"}.get(st.value, "octree read failed"))
            return
        self._h = L.orc_accel_build(_p(self.verts), self.verts.shape[0], {"AABB": 1, "OCTREE": 2}[kind], int(param))
        if not self._h:
            raise RuntimeError("octree split recursion does not end (the reference overflows its stack here)")

    @classmethod
    def from_octree_file(cls, verts, path):
        return cls(verts, "OCTREE", 0, octree_file=path)

    def write_octree(self, path):
        """cOctree::WriteOctree (src/geometry/cAccelStruct.cpp:485-520): the text cache the reference
        keeps beside the mesh."""
        if lib().orc_octree_write(self._h, os.fsencode(path)) != 0:
            raise RuntimeError("cannot write %s (not an octree, or the file cannot be opened)" % path)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_accel_free(self._h)
            self._h = None

    def info(self):
        out = np.zeros(5, dtype=np.int64)
        lib().orc_accel_info(self._h, _p(out))
        return {"boxes": int(out[0]), "face_refs": int(out[1]), "stranded_refs": int(out[2]),
                "unreachable_faces": int(out[3]), "reference_would_exit": int(out[4])}

    def raycast(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float64).reshape(-1, 3)
        ids = np.empty(o.shape[0], dtype=np.int32)
        pts = np.empty((o.shape[0], 3), dtype=np.float64)
        lib().orc_accel_raycast_many(self._h, o.shape[0], _p(o), _p(d), _p(ids), _p(pts))
        return ids, pts

    def vistest(self, p1, p2):
        p1 = np.ascontiguousarray(p1, dtype=np.float64).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, dtype=np.float64).reshape(-1, 3)
        vis = np.empty(p1.shape[0], dtype=np.int32)
        lib().orc_accel_vistest_many(self._h, p1.shape[0], _p(p1), _p(p2), _p(vis))
        return vis.astype(bool)
