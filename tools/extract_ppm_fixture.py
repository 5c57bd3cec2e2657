#!/usr/bin/env python3
"""Writes tests/golden/mis_ppm_stats.json from the one raw output image the reference holds
(results/mis.ppm): file hash and size, image statistics, and the first 24 pixels with the exact
bytes that encode them.   usage: extract_ppm_fixture.py [/root/reference]"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pathtracer_amd.api import read_ppm          # noqa: E402
from pathtracer_amd.image import image_stats     # noqa: E402

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
path = os.path.join(ref, "results/mis.ppm")
raw = open(path, "rb").read()
img = read_ppm(path)
head = raw.split(b"\n", 3)
n = 24
body = head[3].split(b" ")
fix = {"source": "results/mis.ppm", "sha256": hashlib.sha256(raw).hexdigest(), "bytes": len(raw),
       "stats": image_stats(img),
       "header": (b"\n".join(head[:3]) + b"\n").decode(),
       "first_pixels": img.reshape(-1, 3)[:n].tolist(),
       "first_pixels_text": (b" ".join(body[:3 * n]) + b" ").decode(),
       "ends_with": raw[-8:].decode(), "newlines": raw.count(b"\n")}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "../tests/golden/mis_ppm_stats.json")
json.dump(fix, open(out, "w"), indent=1)
print(out, fix["stats"])
