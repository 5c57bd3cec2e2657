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
A shading or queueing bug on a path that merely crosses a tie therefore still fails.

The second named exception is AMPLIFICATION.  The bar is absolute, and a path over curved, smoothly
shaded geometry multiplies an error of one of its directions at every further bounce (room.obj at
3840 x 2160: the GPU and the oracle agree to 1e-15 after one bounce of such a path, 1e-13 after three,
3e-9 after five -- a hundredfold per bounce; the reference's own acos-based hemisphere sample is
itself only good to 1e-14 near the pole).  The oracle measures that factor on the sample itself
(Oracle.amplification: the value's reaction to turning the direction drawn at bounce b by 1e-9
radians, largest over b), and a sample without ties passes if its difference is at most
AMPLIFIED_ULPS double-precision ulps times that factor -- below the bar itself wherever the factor
is below 4 400, which is all but a few samples in a million."""
import numpy as np

F64_TOL = 1e-9
TIE_GAP = 1e-10
MAX_TIES = 6
AMPLIFIED_ULPS = 1024   # direction error, in ulps of 1.0, a difference may correspond to
ULP = 2.220446049250313e-16


def bar_of(rgb, tol=F64_TOL):
    """F64_TOL absolute for values up to 1 (every clamped sample), relative beyond (the physically
    based estimator does not clamp)."""
    return tol * max(1.0, float(np.abs(rgb).max()))


def match_sample(gpu_rgb, O, x, y, s, tol=F64_TOL, gap=TIE_GAP):
    """One sample of pixel (column x, row y) as the GPU computed it alone against the oracle.
    Returns (status, mask, detail): "equal" -- agrees with the oracle's own path; "tie" -- agrees
    with the oracle's path under tie mask `mask` (non-zero); "amplified" -- no tie on the path, and the
    difference is what AMPLIFIED_ULPS ulps of direction error become on this path (detail["amplification"]);
    "unmatched" -- none of these."""
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
    amp = O.amplification(x, y, s) if nt == 0 else None
    if amp is not None and d0 <= AMPLIFIED_ULPS * ULP * amp:
        return "amplified", 0, dict(ties=0, diff=d0, amplification=amp)
    return "unmatched", 0, dict(ties=nt, diff=d0, nearest_alternative=best[0], faces=faces.tolist(), amplification=amp,
                                smallest_margin=float(margins.min()) if margins.size else float("inf"))


def prove_pixels(ctx, O, pixels, samples, seed=13, render_kw=None, render_row=None, tol=F64_TOL, gap=TIE_GAP,
                 verbose=None):
    """Puts pixels [(row, column)] to the proof: each sample of `samples` is rendered alone on the GPU
    (through the C ABI, one row at a time) and matched with match_sample.  render_row(y, s) -> [W, 3]
    replaces the default ctx.render call (for callers that keep the frame on the device).
    Returns {"pixels", "proven_ties" (pixels whose every differing sample is a named tie alternative),
    "amplified" (pixels without an unmatched sample and without a tie sample whose differing samples are
    amplified roundings; their factors in "amplifications"),
    "tie_samples", "amplified_samples", "silent_ties" (samples that meet a tie and agree all the same), "no_sample_differs"
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
    out = dict(pixels=0, proven_ties=0, amplified=0, tie_samples=0, amplified_samples=0, silent_ties=0, no_sample_differs=0,
               amplifications=[], unmatched=[])
    for y, xs in sorted(by_row.items()):
        ties = {x: 0 for x in xs}
        amps = {x: 0 for x in xs}
        bad = {x: 0 for x in xs}
        for s in samples:
            row = render_row(y, int(s))
            for x in xs:
                status, mask, detail = match_sample(row[x], O, x, y, int(s), tol, gap)
                if status == "tie":
                    ties[x] += 1
                    out["tie_samples"] += 1
                elif status == "amplified":
                    amps[x] += 1
                    out["amplified_samples"] += 1
                    out["amplifications"].append(detail["amplification"])
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
            elif bad[x] == 0 and amps[x] > 0:
                out["amplified"] += 1
            elif bad[x] == 0:
                out["no_sample_differs"] += 1
    return out


def outliers(img, oimg, tol=F64_TOL, relative=False):
    """Pixels [(row, column)] of two [H, W, 3] sums that differ by more than the bar."""
    img, oimg = np.asarray(img), np.asarray(oimg)
    d = np.abs(img - oimg)
    d[np.isnan(img) & np.isnan(oimg)] = 0.0      # not a number in both: the same
    d[np.isnan(d)] = np.inf                      # in one of them: a difference no bar covers
    diff = d.max(axis=-1)
    bar = tol * np.maximum(1.0, np.abs(oimg).max(axis=-1)) if relative else tol
    return np.argwhere(diff > bar), diff


def check_f64_image(img, oimg, ctx, O, samples, seed=13, origin=(0, 0), render_kw=None, relative=False, max_pixels=None):
    """Compares two f64 sums and proves every outlier.  Returns {"pixels", "beyond", "proven_ties",
    "amplified", "explained" (their sum), "unmatched": [...]} and never raises: the caller decides
    (tests assert `not unmatched` and `explained == put_to_the_proof`)."""
    bad, diff = outliers(img, oimg, relative=relative)
    rec = dict(pixels=int(diff.size), beyond=int(len(bad)), largest_difference=float(diff.max()) if diff.size else 0.0,
               proven_ties=0, amplified=0, explained=0, amplifications=[], unmatched=[], no_sample_differs=0, put_to_the_proof=0)
    if len(bad):
        pick = bad if max_pixels is None or len(bad) <= max_pixels else bad[np.random.default_rng(13).permutation(len(bad))[:max_pixels]]
        r = prove_pixels(ctx, O, [(int(y) + origin[0], int(x) + origin[1]) for y, x in pick], samples, seed=seed, render_kw=render_kw)
        rec.update(proven_ties=r["proven_ties"], amplified=r["amplified"], explained=r["proven_ties"] + r["amplified"],
                   amplifications=r["amplifications"], unmatched=r["unmatched"], no_sample_differs=r["no_sample_differs"],
                   put_to_the_proof=int(len(pick)))
    return rec
