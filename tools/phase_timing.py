"""Where the waves of the ring tracer spend their cycles: renders one frame with the diagnostic build
of the library (make -C pathtracer_amd/csrc PHASE=1), whose waves add up the shader cycles of each
part of their loop; a few of them print their totals (device printf, on stdout).

    python tools/phase_timing.py config/room.conf [--spp 16] [--mode f64] [--estimator fast]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PTX_LIB_OVERRIDE", os.path.join(ROOT, "pathtracer_amd", "libptx_amd_phase.so"))

import pathtracer_amd as pt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("conf")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--mode", default="f64")
    ap.add_argument("--estimator", default="reference")
    a = ap.parse_args()
    s = pt.Scene.load_conf(os.path.join(ROOT, a.conf), root_dir=ROOT, width=a.width, height=a.height, spp=a.spp)
    c = pt.Context(0)
    c.build_accel(s)
    _, st = c.render(mode=a.mode, estimator=a.estimator)
    print(c.last_kernel(), {k: st[k] for k in ("paths", "segments", "ms_trace", "wave_iterations", "bvh_nodes", "bvh_tris")}, flush=True)


if __name__ == "__main__":
    main()
