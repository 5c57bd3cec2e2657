"""The largest configuration of the baseline at its full size on one GPU: config/room.conf at
3840 x 2160 with 4096 samples per pixel (3.4e10 paths), rendered three ways -- in one piece, as the
eight interleaved row-band shards an 8-GPU run cuts it into (bands of 8 rows, one phase per rank),
and as eight sample ranges of 512 accumulated in rank order -- which must give the same sums bit
for bit.  Prints one JSON line (times, rates, a digest of the sums) and writes the PPM.

    python tools/full_config_room4k.py [--spp 4096] [--out room4k.ppm]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracer_amd as pt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=4096)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    W, H = 3840, 2160
    s = pt.Scene.load_conf(os.path.join(ROOT, "config/room.conf"), root_dir=ROOT, width=W, height=H, spp=a.spp)
    c = pt.Context(0)
    c.build_accel(s)
    rec = {"config": "config/room.conf at %d x %d, %d spp, f64" % (W, H, a.spp), "paths": W * H * a.spp, "ranks_replayed": a.ranks}
    t0 = time.time()
    full, st = c.render(mode="f64", spp=a.spp)
    dt = time.time() - t0
    rec["one_piece"] = {"seconds": round(dt, 2), "mpaths_per_s": round(st["paths"] / dt / 1e6, 1),   # wall time, the copy of the frame included
                        "segments": int(st["segments"]), "kernel": c.last_kernel()}
    print("one piece: %s" % rec["one_piece"], file=sys.stderr, flush=True)
    rec["sha256_of_sums"] = hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest()
    t0 = time.time()
    acc = np.zeros_like(full)
    segs = 0
    for k in range(a.ranks):
        part, sp = c.render(mode="f64", spp=a.spp, row_stride=a.ranks, row_phase=k, band=8)
        acc += part
        segs += sp["segments"]
        print("band shard %d of %d done" % (k + 1, a.ranks), file=sys.stderr, flush=True)
    rec["row_bands"] = {"seconds": round(time.time() - t0, 2), "equal_bit_for_bit": bool(np.array_equal(acc, full)), "segments_add_up": bool(segs == st["segments"])}
    del acc
    t0 = time.time()
    per = a.spp // a.ranks
    run = np.zeros_like(full)
    for k in range(a.ranks):
        run, _ = c.render(mode="f64", spp=per, sample_offset=k * per, spp_total=a.spp, accumulate=True, out=run)
        print("sample range %d of %d done" % (k + 1, a.ranks), file=sys.stderr, flush=True)
    rec["sample_ranges"] = {"seconds": round(time.time() - t0, 2), "equal_bit_for_bit": bool(np.array_equal(run, full))}
    img = pt.finalize_image(full)
    rec["image"] = {"mean": float(img.mean()), "max": float(img.max()), "black_pixels": int((img.max(axis=-1) == 0).sum())}
    if a.out:
        pt.write_ppm(a.out, img)
        with open(a.out, "rb") as f:
            rec["ppm"] = {"path": a.out, "bytes": os.path.getsize(a.out), "sha256": hashlib.sha256(f.read()).hexdigest()}
    print(json.dumps(rec), flush=True)
    return 0 if rec["row_bands"]["equal_bit_for_bit"] and rec["sample_ranges"]["equal_bit_for_bit"] else 1


if __name__ == "__main__":
    sys.exit(main())
