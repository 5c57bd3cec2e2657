"""Image tooling around the reference's output format (ASCII P3 PPM, `int(c*255)` per channel, no
gamma: src/tracing/cPathTracer.cpp:261-284 of the reference): statistics of one image, comparison of
two, and merging of partial radiance sums (row bands or sample ranges rendered separately).

Command line:  python -m pathtracer_amd.image A.ppm [B.ppm] [--tolerance T]
prints one JSON object; with two images the exit code is 1 when the relative MSE exceeds T.
"""
import json
import sys

import numpy as np

from .api import read_ppm


def to_linear(img):
    """Pixel values in [0, 1] from 8-bit PPM values (bin centres: the writer truncates)."""
    a = np.asarray(img)
    if np.issubdtype(a.dtype, np.integer):
        return (a.astype(np.float64) + 0.5) / 255.0
    return a.astype(np.float64)


def relative_mse(a, b):
    """mean over pixels of sum over channels of (a-b)^2 / (b^2 + 1e-2); `b` is the reference image."""
    a, b = to_linear(a), to_linear(b)
    if a.shape != b.shape:
        raise ValueError("image shapes differ: %s vs %s" % (a.shape, b.shape))
    return float(np.mean(np.sum((a - b) ** 2 / (b ** 2 + 1e-2), axis=-1)))


def image_stats(img):
    """Size, mean RGB (0..255 scale) and the fraction of saturated channel values."""
    a = np.asarray(img)
    v = a.astype(np.float64) if np.issubdtype(a.dtype, np.integer) else np.floor(a.astype(np.float64) * 255.0)
    return {"height": int(a.shape[0]), "width": int(a.shape[1]),
            "mean_rgb": [float(x) for x in v.reshape(-1, 3).mean(0)],
            "saturated_fraction": float((v >= 255).mean()),
            "black_pixels": int((v.reshape(-1, 3).max(1) == 0).sum())}


def compare(a, b):
    """Relative MSE, PSNR over the 8-bit range, largest channel difference and where it is."""
    la, lb = to_linear(a), to_linear(b)
    d = np.abs(la - lb)
    at = np.unravel_index(int(np.argmax(d.max(-1))), d.shape[:2])
    mse = float(np.mean((la - lb) ** 2))
    return {"relative_mse": relative_mse(a, b), "mse": mse,
            "psnr_db": float("inf") if mse == 0.0 else float(-10.0 * np.log10(mse)),
            "max_abs_diff": float(d.max()), "max_at_row_col": [int(at[0]), int(at[1])],
            "differing_pixels": int((d.max(-1) > 0).sum())}


def merge_partial_sums(parts):
    """Sum of partial radiance images (each the pre-final-clamp sums of some rows or of some sample
    range at the same spp_total, zeros elsewhere), added in the order given.  The per-pixel value is
    linear in the samples and the final clamp comes afterwards (cPathTracer.cpp:692,701-702)."""
    out = np.zeros_like(np.asarray(parts[0], dtype=np.float64))
    for p in parts:
        p = np.asarray(p, dtype=np.float64)
        if p.shape != out.shape:
            raise ValueError("partial images differ in shape")
        out += p
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    tol = None
    if "--tolerance" in argv:
        i = argv.index("--tolerance")
        tol = float(argv[i + 1])
        del argv[i:i + 2]
    if len(argv) not in (1, 2):
        print("usage: python -m pathtracer_amd.image A.ppm [B.ppm] [--tolerance T]")
        return 2
    a = read_ppm(argv[0])
    out = {"a": dict(path=argv[0], **image_stats(a))}
    rc = 0
    if len(argv) == 2:
        b = read_ppm(argv[1])
        out["b"] = dict(path=argv[1], **image_stats(b))
        out["compare"] = compare(a, b)
        if tol is not None:
            out["tolerance"] = tol
            rc = int(out["compare"]["relative_mse"] > tol)
    print(json.dumps(out))
    return rc


if __name__ == "__main__":
    sys.exit(main())
