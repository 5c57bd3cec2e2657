"""The f64 image-parity rule, in one place (TEST INFRASTRUCTURE ONLY: tests/, tools/, the smoke check
and the checks of bench.py use it; the product never imports it).

An f64 render runs the reference's arithmetic on the oracle's random numbers, so the per-pixel sums
agree to F64_TOL -- except on a path that meets a *tie*: a closest-hit search whose winner and
runner-up lie within TIE_GAP of each other (the Cornell box asset has two coplanar, oppositely wound
quads in its ceiling).  Which of the two a renderer reports hinges on the last ulp of the ray
direction, where the device's and glibc's sin / cos / acos may differ; the reference itself decides
it by strict '<' over ascending face ids (cAccelStruct.cpp:39-51).  No anonymous exception is made
for such paths.  A differing sample passes only if the value the GPU computed for it ALONE equals,
to F64_TOL, the value the oracle computes when it re-traces the sample with the ties on its path
falling another, *named* way (Oracle.tie_alternatives: bit k of a mask hands tie k to the runner-up).
A shading or queueing bug on a path that merely crosses a tie therefore still fails."""
import numpy as np

F64_TOL = 1e-9
TIE_GAP = 1e-10
MAX_TIES = 6


def bar_of(rgb, tol=F64_TOL):
    """F64_TOL absolute for values up to 1 (every clamped sample), relative beyond (the physically
    based estimator does not clamp)."""
    return tol * max(1.0, float(np.abs(rgb).max()))


def match_sample(gpu_rgb, O, x, y, s, tol=F64_TOL, gap=TIE_GAP):
    """One sample of pixel (column x, row y) as the GPU computed it alone against the oracle.
    Returns (status, mask, detail): "equal" -- agrees with the oracle's own path; "tie" -- agrees
    with the oracle's path under tie mask `mask` (non-zero); "unmatched" -- agrees with none."""
    rgb0, faces, margins, nt = O.trace_sample_ties(x, y, s, 0, gap)
    d0 = float(np.abs(gpu_rgb - rgb0).max())
    if d0 <= bar_of(rgb0, tol):
        return "equal", 0, dict(ties=nt, diff=d0)
    best = (np.inf, 0)
    if nt:
        for mask, rgb, _ in O.tie_alternatives(x, y, s, gap=gap, max_ties=MAX_TIES):
            d = float(np.abs(gpu_rgb - rgb).max())
            if mask and d <= bar_of(rgb, tol):
                return "tie", mask, dict(ties=nt, diff=d0)
            best = min(best, (d, mask))
    return "unmatched", 0, dict(ties=nt, diff=d0, nearest_alternative=best[0], faces=faces.tolist(),
                                smallest_margin=float(margins.min()) if margins.size else float("inf"))


def prove_pixels(ctx, O, pixels, samples, seed=13, render_kw=None, render_row=None, tol=F64_TOL, gap=TIE_GAP,
                 verbose=None):
    """Puts pixels [(row, column)] to the proof: each sample of `samples` is rendered alone on the GPU
    (through the C ABI, one row at a time) and matched with match_sample.  render_row(y, s) -> [W, 3]
    replaces the default ctx.render call (for callers that keep the frame on the device).
    Returns {"pixels", "proven_ties" (pixels whose every differing sample is a named tie alternative),
    "tie_samples", "silent_ties" (samples that meet a tie and agree all the same), "no_sample_differs"
    (pixels none of whose samples differs: the sums differ by accumulated rounding only),
    "unmatched": [(row, column, sample, detail)]}."""
    kw = dict(render_kw or {})
    if render_row is None:
        def render_row(y, s):
            one, _ = ctx.render(mode="f64", spp=1, sample_offset=s, spp_total=1, rows=(y, y + 1), seed=int(seed), **kw)
            return one[y]
    by_row = {}
    for y, x in pixels:
        by_row.setdefault(int(y), []).append(int(x))
    out = dict(pixels=0, proven_ties=0, tie_samples=0, silent_ties=0, no_sample_differs=0, unmatched=[])
    for y, xs in sorted(by_row.items()):
        ties = {x: 0 for x in xs}
        bad = {x: 0 for x in xs}
        for s in samples:
            row = render_row(y, int(s))
            for x in xs:
                status, mask, detail = match_sample(row[x], O, x, y, int(s), tol, gap)
                if status == "tie":
                    ties[x] += 1
                    out["tie_samples"] += 1
                elif status == "unmatched":
                    bad[x] += 1
                    out["unmatched"].append((y, x, int(s), detail))
                elif detail["ties"]:
                    out["silent_ties"] += 1
                if verbose and status != "equal":
                    verbose("pixel (%d, %d) sample %d: %s (mask %#x) %s" % (y, x, s, status, mask, detail))
        for x in xs:
            out["pixels"] += 1
            if bad[x] == 0 and ties[x] > 0:
                out["proven_ties"] += 1
            elif bad[x] == 0:
                out["no_sample_differs"] += 1
    return out


def outliers(img, oimg, tol=F64_TOL, relative=False):
    """Pixels [(row, column)] of two [H, W, 3] sums that differ by more than the bar."""
    img, oimg = np.asarray(img), np.asarray(oimg)
    diff = np.abs(img - oimg).max(axis=-1)
    bar = tol * np.maximum(1.0, np.abs(oimg).max(axis=-1)) if relative else tol
    return np.argwhere(diff > bar), diff


def check_f64_image(img, oimg, ctx, O, samples, seed=13, origin=(0, 0), render_kw=None, relative=False, max_pixels=None):
    """Compares two f64 sums and proves every outlier.  Returns {"pixels", "beyond_1e-9", "proven_ties",
    "unmatched": [...]} and never raises: the caller decides (tests assert `not unmatched`)."""
    bad, diff = outliers(img, oimg, relative=relative)
    rec = dict(pixels=int(diff.size), beyond=int(len(bad)), largest_difference=float(diff.max()) if diff.size else 0.0,
               proven_ties=0, unmatched=[], no_sample_differs=0, put_to_the_proof=0)
    if len(bad):
        pick = bad if max_pixels is None or len(bad) <= max_pixels else bad[np.random.default_rng(13).permutation(len(bad))[:max_pixels]]
        r = prove_pixels(ctx, O, [(int(y) + origin[0], int(x) + origin[1]) for y, x in pick], samples, seed=seed, render_kw=render_kw)
        rec.update(proven_ties=r["proven_ties"], unmatched=r["unmatched"], no_sample_differs=r["no_sample_differs"],
                   put_to_the_proof=int(len(pick)))
    return rec
