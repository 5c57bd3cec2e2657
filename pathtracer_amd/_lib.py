"""ctypes binding of libptx_amd.so (the C ABI declared in include/ptx.h).

The library holds the whole product: scene front end, BVH builder and the HIP kernels.  There is
no Python or CPU fallback -- if the shared object is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTX_LIB_OVERRIDE") or os.path.join(_HERE, "libptx_amd.so")

PTX_OK = 0
PTX_MODE_F64 = 0
PTX_MODE_F32 = 1
PTX_ESTIMATOR_REFERENCE = 0
PTX_ESTIMATOR_FAST = 1


class Overrides(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("depth", C.c_int32), ("model", C.c_char_p), ("full_region", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("n_tris", C.c_int32), ("n_mats", C.c_int32), ("n_lights", C.c_int32),
                ("n_square_lights", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("depth", C.c_int32), ("spp", C.c_int32), ("enable_indirect", C.c_int32),
                ("enable_bary_normal", C.c_int32), ("enable_path_tracing", C.c_int32),
                ("region", C.c_int32 * 4), ("accel_type", C.c_int32),
                ("fov_deg", C.c_double), ("near_plane", C.c_double), ("world_scale", C.c_double)]


PTX_ABI_VERSION = 2


class RenderOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_int32), ("spp", C.c_int32), ("sample_offset", C.c_int32),
                ("spp_total", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
                ("row_stride", C.c_int32), ("row_phase", C.c_int32), ("band", C.c_int32),
                ("depth", C.c_int32), ("seed", C.c_uint64), ("batch_paths", C.c_int64),
                ("accumulate", C.c_int32), ("profile", C.c_int32), ("estimator", C.c_int32),
                ("enqueue_only", C.c_int32)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(RenderOpts)


class Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved_", C.c_uint32), ("paths", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("passes", C.c_uint64), ("wave_iterations", C.c_uint64), ("records", C.c_uint64), ("bvh_nodes", C.c_uint64), ("bvh_tris", C.c_uint64), ("ms_total", C.c_double), ("ms_trace", C.c_double),
                ("ms_accumulate", C.c_double), ("ms_other", C.c_double), ("state_bytes", C.c_uint64)]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(Stats)


# every symbol include/ptx.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
SYMBOLS = {
    "ptx_scene_load_conf": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(Overrides), C.POINTER(_VP)]),
    "ptx_scene_from_arrays": (C.c_int, [_VP, _VP, _VP, C.c_int32, _VP, C.c_int32, _VP, C.c_int32,
                                        _VP, _VP, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "ptx_scene_destroy": (None, [_VP]),
    "ptx_scene_get_info": (C.c_int, [_VP, C.POINTER(SceneInfo)]),
    "ptx_scene_copy_geometry": (C.c_int, [_VP, _VP, _VP, _VP]),
    "ptx_scene_copy_vertex_ids": (C.c_int, [_VP, _VP]),
    "ptx_scene_copy_tables": (C.c_int, [_VP, _VP, _VP]),
    "ptx_scene_copy_camera": (C.c_int, [_VP, _VP, _VP, _VP]),
    "ptx_scene_result_path": (C.c_char_p, [_VP]),
    "ptx_scene_plane_records": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "ptx_scene_bvh_check": (C.c_int, [_VP, C.c_int32, _VP]),
    "ptx_context_create": (C.c_int, [C.c_int32, C.POINTER(_VP)]),
    "ptx_context_set_stream": (C.c_int, [_VP, _VP]),
    "ptx_context_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int64]),
    "ptx_context_destroy": (None, [_VP]),
    "ptx_build_accel": (C.c_int, [_VP, _VP]),
    "ptx_accel_info": (C.c_int, [_VP, _VP]),
    "ptx_generate_rays": (C.c_int, [_VP, _VP, _VP]),
    "ptx_first_hit": (C.c_int, [_VP, _VP, _VP]),
    "ptx_render": (C.c_int, [_VP, C.POINTER(RenderOpts), _VP, C.c_int32, C.POINTER(Stats)]),
    "ptx_render_static": (C.c_int, [_VP, C.POINTER(RenderOpts), _VP, C.POINTER(Stats)]),
    "ptx_context_last_kernel": (C.c_char_p, [_VP]),
    "ptx_render_sharded": (C.c_int, [_VP, C.c_int32, C.POINTER(RenderOpts), C.c_int32, C.c_int32, _VP, _VP]),
    "ptx_raycast": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP]),
    "ptx_vistest": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP]),
    "ptx_finalize_image": (C.c_int, [_VP, C.c_int64, _VP]),
    "ptx_write_ppm": (C.c_int, [C.c_char_p, _VP, C.c_int32, C.c_int32]),
    "ptx_last_error": (C.c_char_p, []),
    "ptx_version": (C.c_char_p, []),
    "ptx_abi_version": (C.c_int32, []),
}

_lib = None


class PtxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ptx error %d: %s" % (code, message))
        self.code = code
        self.message = message


def lib():
    """Loads libptx_amd.so (raises if it has not been built: `make -C pathtracer_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("pathtracer_amd/libptx_amd.so is missing -- build it with "
                               "`make -C pathtracer_amd/csrc` (hipcc, gfx950); there is no fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.ptx_abi_version() != PTX_ABI_VERSION:
            raise RuntimeError("libptx_amd.so speaks ABI version %d, this binding %d: rebuild the library"
                               % (L.ptx_abi_version(), PTX_ABI_VERSION))
        _lib = L
    return _lib


def check(code):
    if code != PTX_OK:
        raise PtxError(code, lib().ptx_last_error().decode("utf-8", "replace"))
