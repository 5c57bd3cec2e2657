"""Image tooling around the reference's output format (ASCII P3 PPM, `int(c*255)` per channel, no
gamma: src/tracing/cPathTracer.cpp:261-284 of the reference): statistics of one image, comparison of
two, and merging of partial radiance sums (row bands or sample ranges rendered separately).

Command line:  python -m pathtracer_amd.image A.ppm [B.ppm] [--tolerance T]
prints one JSON object (relative MSE -- the measure of the parity bar --, PSNR, CIE76 colour
difference, a FLIP-like perceptual mean); with two images the exit code is 1 when the relative MSE
exceeds T.

               python -m pathtracer_amd.image --merge A.part B.part ... -o out.ppm
adds parts of one frame written by `pathtracer <conf> --dump-partial FILE [--rows A:B] [--samples A:B]`
(the pre-clamp sums of some rows and some samples: a long render cut into pieces, done piece by piece,
resumed where it stopped), applies the final clamp and writes the reference's PPM.  Parts that
overlap, disagree about the frame (size, sample count, depth, seed, precision, estimator) or do not
add up to the whole frame are refused.
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


def to_lab(img):
    """CIE L*a*b* (D65) of an image whose values are LINEAR radiance in [0, 1] -- what the reference
    writes: its PPM has no gamma.  Values above 1 are clipped."""
    rgb = np.clip(to_linear(img), 0.0, 1.0)
    m = np.array([[0.4124564, 0.3575761, 0.1804375], [0.2126729, 0.7151522, 0.0721750], [0.0193339, 0.1191920, 0.9503041]])
    xyz = rgb @ m.T / np.array([0.95047, 1.0, 1.08883])
    f = np.where(xyz > (6.0 / 29.0) ** 3, np.cbrt(xyz), xyz / (3 * (6.0 / 29.0) ** 2) + 4.0 / 29.0)
    return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], axis=-1)


def delta_e(a, b):
    """Per-pixel colour difference CIE76 (Euclidean distance in L*a*b*): about 2.3 is a just
    noticeable difference."""
    return np.sqrt(np.sum((to_lab(a) - to_lab(b)) ** 2, axis=-1))


def flip_lite(a, b, radius=2):
    """A perceptual error map in [0, 1] in the spirit of FLIP, much reduced: the CIE76 difference of
    the two images after a small box blur (a stand-in for the eye's spatial filtering, so that
    per-pixel Monte Carlo noise weighs less than structure), mapped through 1 - exp(-dE / 20).
    No viewing-distance model, no feature term: a ranking aid, not the published metric."""
    def blur(x):
        x = to_linear(x)
        k = 2 * radius + 1
        p = np.pad(x, ((radius, radius), (radius, radius), (0, 0)), mode="edge")
        c = np.cumsum(np.cumsum(p, axis=0), axis=1)
        c = np.pad(c, ((1, 0), (1, 0), (0, 0)))
        return (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
    return 1.0 - np.exp(-delta_e(blur(a), blur(b)) / 20.0)


def compare(a, b):
    """Relative MSE, PSNR over the 8-bit range, largest channel difference and where it is, the
    CIE76 colour difference (mean, 95th percentile, largest) and the mean of the FLIP-like map."""
    la, lb = to_linear(a), to_linear(b)
    d = np.abs(la - lb)
    at = np.unravel_index(int(np.argmax(d.max(-1))), d.shape[:2])
    mse = float(np.mean((la - lb) ** 2))
    de = delta_e(a, b)
    return {"relative_mse": relative_mse(a, b), "mse": mse,
            "psnr_db": float("inf") if mse == 0.0 else float(-10.0 * np.log10(mse)),
            "max_abs_diff": float(d.max()), "max_at_row_col": [int(at[0]), int(at[1])],
            "differing_pixels": int((d.max(-1) > 0).sum()),
            "delta_e76_mean": float(de.mean()), "delta_e76_p95": float(np.percentile(de, 95)), "delta_e76_max": float(de.max()),
            "flip_lite_mean": float(flip_lite(a, b).mean())}


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


PART_MAGIC = b"PTXPART1"


def write_part(path, rows_sum, width, height, spp_total, rows, samples, mode=0, estimator=0, depth=0, seed=13):
    """A part file as the command line writes it: rows_sum = [rows[1] - rows[0], width, 3] float64."""
    a = np.ascontiguousarray(rows_sum, dtype="<f8")
    if a.shape != (rows[1] - rows[0], width, 3):
        raise ValueError("rows_sum has shape %s, the header says %s" % (a.shape, (rows[1] - rows[0], width, 3)))
    head = np.array([width, height, spp_total, rows[0], rows[1], samples[0], samples[1], mode, estimator, depth, 0, 0], dtype="<i4")
    with open(path, "wb") as f:
        f.write(PART_MAGIC)
        f.write(head.tobytes())
        f.write(np.array([seed], dtype="<u8").tobytes())
        f.write(a.tobytes())


def read_part(path):
    """(header dict, rows_sum [n_rows, width, 3]) of a part file; raises ValueError on a malformed one."""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 64 or blob[:8] != PART_MAGIC:
        raise ValueError("%s is not a partial-sum file" % path)
    h = np.frombuffer(blob, dtype="<i4", count=12, offset=8)
    head = dict(width=int(h[0]), height=int(h[1]), spp_total=int(h[2]), rows=(int(h[3]), int(h[4])), samples=(int(h[5]), int(h[6])),
                mode=int(h[7]), estimator=int(h[8]), depth=int(h[9]), seed=int(np.frombuffer(blob, dtype="<u8", count=1, offset=56)[0]))
    r0, r1 = head["rows"]
    s0, s1 = head["samples"]
    if not (0 <= r0 < r1 <= head["height"] and 0 <= s0 < s1 <= head["spp_total"] and head["width"] > 0):
        raise ValueError("%s: ranges outside the frame it names" % path)
    n = (r1 - r0) * head["width"] * 3
    if len(blob) != 64 + 8 * n:
        raise ValueError("%s: %d bytes of sums, the header promises %d" % (path, len(blob) - 64, 8 * n))
    return head, np.frombuffer(blob, dtype="<f8", count=n, offset=64).reshape(r1 - r0, head["width"], 3)


def merge_parts(paths):
    """The whole frame's pre-clamp sums [H, W, 3] from part files.  Every (row, sample) of the frame must
    be covered exactly once: parts are rectangles of rows x samples, and they must agree about the frame.
    Parts of the same rows are added in ascending sample order (the order a single render adds them in)."""
    parts = [read_part(p) + (p,) for p in paths]
    if not parts:
        raise ValueError("no parts")
    first = parts[0][0]
    for head, _, path in parts:
        for k in ("width", "height", "spp_total", "mode", "estimator", "depth", "seed"):
            if head[k] != first[k]:
                raise ValueError("%s: %s is %s, %s says %s" % (path, k, head[k], parts[0][2], first[k]))
    H, W, S = first["height"], first["width"], first["spp_total"]
    cover = np.zeros((H, S), dtype=np.int32)                    # how many parts hold (row, sample)
    for head, _, path in parts:
        (r0, r1), (s0, s1) = head["rows"], head["samples"]
        if cover[r0:r1, s0:s1].any():
            raise ValueError("%s overlaps another part (rows %d:%d, samples %d:%d)" % (path, r0, r1, s0, s1))
        cover[r0:r1, s0:s1] += 1
    if not cover.all():
        r, s = np.argwhere(cover == 0)[0]
        raise ValueError("the parts do not cover the frame: row %d, sample %d is in none of them" % (r, s))
    out = np.zeros((H, W, 3))
    for head, rows_sum, _ in sorted(parts, key=lambda t: t[0]["samples"][0]):
        out[head["rows"][0]:head["rows"][1]] += rows_sum
    return out, first


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "--merge":
        if "-o" not in argv or argv.index("-o") + 1 >= len(argv):
            print("usage: python -m pathtracer_amd.image --merge A.part B.part ... -o out.ppm")
            return 2
        i = argv.index("-o")
        out_path, paths = argv[i + 1], argv[1:i] + argv[i + 2:]
        from .api import finalize_image, write_ppm
        try:
            total, head = merge_parts(paths)
        except (ValueError, OSError) as e:
            print("[error] %s" % e)
            return 1
        write_ppm(out_path, finalize_image(total))
        print(json.dumps({"merged": len(paths), "width": head["width"], "height": head["height"], "spp_total": head["spp_total"], "out": out_path}))
        return 0
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
