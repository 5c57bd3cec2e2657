"""The reference's results/cbox_cube.png against the CPU oracle rendering config/cbox_cube_ref.conf,
for a scan of the tracer depth (the screenshot's conf revision is unknown; nothing else is varied).
Uses the committed fixture (tests/golden/cbox_cube_ref64.npz), so it runs anywhere.  Prints one JSON
document: per depth, the ratio of the per-channel means (oracle / screenshot) and the RMS difference
of the 64 x 64 pooled frames, inside and outside the glass cube.

    python tools/reference_image_report.py [--spp 32] > profiles/r03_cbox_cube_reference_image.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle, scene_io  # noqa: E402
import test_reference_image as tri  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=32)
    a = ap.parse_args()
    golden = os.path.join(ROOT, "tests", "golden")
    g, valid, hit, lit = tri.fixture(golden)
    sc = scene_io.load_conf(os.path.join(ROOT, tri.CONF), root_dir=ROOT)
    ids = oracle.Oracle(sc).first_hit().reshape(512, 512)
    rec = {"picture": "results/cbox_cube.png of the reference (sha256 %s), 2x screenshot of a 512 x 512 render, display profile undone" % bytes(g["sha256"]).hex(),
           "conf": tri.CONF, "spp": a.spp, "screenshot_shows_render_rows_cols": [int(v) for v in g["valid"]],
           "pixels_hit_in_screenshot": int(hit.sum()), "hit_mask_mismatches_vs_oracle_first_hit": int(((ids >= 0) != hit)[valid].sum()),
           "emitter_pixels_in_screenshot": int(lit.sum()),
           "emitter_mask_mismatches_vs_oracle_first_hit": int((((ids == 10) | (ids == 11)) != lit)[valid].sum()),
           "depth_scan": {}}
    for depth in (3, 5, 7, 10, 15):
        sc = scene_io.load_conf(os.path.join(ROOT, tri.CONF), root_dir=ROOT, depth=depth)
        r = tri.compare(tri.pooled(oracle.Oracle(sc, seed=13).render(spp=a.spp), oracle.finalize), g)
        rec["depth_scan"][str(depth)] = {k: {"mean_ratio_rgb": [round(float(x), 4) for x in v[0]], "rms": round(v[1], 5)} for k, v in r.items()}
        print("depth %d done" % depth, file=sys.stderr, flush=True)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
