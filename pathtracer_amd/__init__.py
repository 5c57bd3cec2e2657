"""MI355X-native wavefront path-tracing core behind the drop-in surface of Rullec/PathTracer
(conf files, OBJ/MTL, camera block, P3 PPM).  All computation lives in libptx_amd.so (C ABI in
include/ptx.h, HIP kernels for gfx950); this package is the thin host-side mirror."""
from .api import (Context, PathTracer, PtxError, Scene, finalize_image, read_ppm,  # noqa: F401
                  render_sharded, write_ppm)
from ._lib import LIB_PATH, PTX_MODE_F32, PTX_MODE_F64, SYMBOLS, lib  # noqa: F401

__version__ = "0.1"
