"""Render function for `bench.py --rehearse bench_rehearsal` (tests only): the CPU oracle in the
place of the GPU core, so that bench.py's own launch (N ranks started by `python bench.py --gpus
N`), sharding, reduce, timing and result line can be exercised over gloo on a machine without a
GPU.  The line such a run prints is marked as a rehearsal: it measures nothing."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def setup(args, rank, world, W, H):
    from oracle import oracle, scene_io

    sc = scene_io.load_conf(os.path.join(ROOT, args.conf), root_dir=ROOT, width=W, height=H, spp=args.spp)
    O = oracle.Oracle(sc, seed=13)

    def render_fn(out, spp, sample_offset, spp_total, row_stride=1, row_phase=0, band=1, **_):
        img = O.render(spp=spp, sample_offset=sample_offset, spp_total=spp_total, threads=1)
        if row_stride > 1:
            rows = np.arange(img.shape[0])
            img[(rows // band) % row_stride != row_phase] = 0
        out.copy_(torch.from_numpy(img))
        return None

    return render_fn, torch.zeros((H, W, 3), dtype=torch.float64)
