"""Makes tests/golden/cbox_cube_ref64.npz from the one picture the reference holds of a scene whose
assets are in its checkout: results/cbox_cube.png ("My own basic scene", README.md:36), a window
screenshot of the render of config/cbox_cube.conf (model/cbox/cbox_test_1.obj).  Runs in the build
container only (the reference is not on the GPU box); the fixture is data -- pooled pixel values and
two bit masks -- not source.

What the file is, measured here: 1024 x 1014 RGBA at 144 dpi with an embedded display ICC profile,
i.e. a 2x ("retina") capture: every value fills an aligned 2 x 2 block, so the picture is 512 x 507
of the 512 x 512 render; 2 columns of grey window chrome on the right.  Registration against the
render was found by sliding it over the first-hit map of this repository's restatement: at the shift
(2, 2) -- screenshot pixel (r, c) shows render pixel (r + 2, c + 2) -- the set of non-black pixels
and the set of saturated (emitter) pixels both equal the restatement's to the pixel, and at no other
shift.  The viewer that was photographed converted the PPM's values to the display's colour space;
the embedded profile is used to convert them back to sRGB values (littleCMS through PIL.ImageCms),
both versions are stored.

Stored (render coordinates, row 0 at the top):
  rgb64      [64][64][3] mean of value / 255 over each 8 x 8 block, profile undone
  rgb64_raw  the same from the raw screenshot values
  weight64   [64][64] share of a block's pixels the screenshot shows
  hit_bits / lit_bits  np.packbits of the 512 x 512 masks "non-black" and "all channels >= 250"
  valid      [row0, col0, row1, col1): the part of the render the screenshot shows
  sha256     of the PNG

    python tools/extract_cbox_cube_fixture.py [/root/reference]
"""
import hashlib
import io
import os
import sys

import numpy as np
from PIL import Image, ImageCms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIFT = (2, 2)


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    path = os.path.join(ref_root, "results", "cbox_cube.png")
    blob = open(path, "rb").read()
    im = Image.open(io.BytesIO(blob))
    assert im.size == (1024, 1014), im.size
    rgb = im.convert("RGB")
    raw = np.asarray(rgb)
    assert (raw[0::2] == raw[1::2]).all() and (raw[:, 0::2] == raw[:, 1::2]).all(), "not an aligned 2x capture"
    src = ImageCms.ImageCmsProfile(io.BytesIO(im.info["icc_profile"]))
    undone = np.asarray(ImageCms.profileToProfile(rgb, src, ImageCms.createProfile("sRGB"), outputMode="RGB"))
    chrome = 2                                                  # grey columns on the right (at 512 scale)

    def place(a):
        a = a[0::2, 0::2][:, :512 - 2 * chrome].astype(np.float64)
        full = np.zeros((512, 512, 3))
        r0, c0 = SHIFT
        h, w = min(a.shape[0], 512 - r0), min(a.shape[1], 512 - c0)
        full[r0:r0 + h, c0:c0 + w] = a[:h, :w]
        return full, (r0, c0, r0 + h, c0 + w)

    full, valid = place(undone)
    full_raw, _ = place(raw)
    mask = np.zeros((512, 512), bool)
    mask[valid[0]:valid[2], valid[1]:valid[3]] = True

    def pool(x):
        return x.reshape(64, 8, 64, 8, -1).mean(axis=(1, 3))

    w64 = pool(mask[..., None].astype(np.float64))[..., 0]
    div = np.maximum(w64, 1e-9)[..., None]
    out = os.path.join(ROOT, "tests", "golden", "cbox_cube_ref64.npz")
    np.savez_compressed(
        out,
        rgb64=(pool(full / 255.0 * mask[..., None]) / div).astype(np.float32),
        rgb64_raw=(pool(full_raw / 255.0 * mask[..., None]) / div).astype(np.float32),
        weight64=w64.astype(np.float32),
        hit_bits=np.packbits((full_raw.sum(axis=-1) > 0) & mask),
        lit_bits=np.packbits((full_raw.min(axis=-1) >= 250) & mask),
        valid=np.asarray(valid, dtype=np.int32),
        shift=np.asarray(SHIFT, dtype=np.int32),
        sha256=np.frombuffer(hashlib.sha256(blob).digest(), dtype=np.uint8),
    )
    print("wrote %s (%d bytes): shows render rows %d..%d, columns %d..%d; %d non-black and %d saturated pixels" % (
        out, os.path.getsize(out), valid[0], valid[2] - 1, valid[1], valid[3] - 1,
        int(((full_raw.sum(axis=-1) > 0) & mask).sum()), int(((full_raw.min(axis=-1) >= 250) & mask).sum())))


if __name__ == "__main__":
    main()
