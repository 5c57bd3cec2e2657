#!/usr/bin/env python3
"""Condenses the reference's model/cbox/*.obj.edge caches (JSON written by
src/geometry/cObjMesh.cpp:158-195 of the reference) into tests/golden/cbox_edge_tables.json.

The .edge tables are the only reference-owned artefacts that pin primitive ids: they record, per
face id, the three undirected vertex-id edges in winding order, i.e. tinyobjloader's fan
triangulation and the running face numbering.  Only the numbers are kept (edge endpoints and the
three edge ids of every face); run from the repository root with the reference mounted."""
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = {}
for name in ("cbox.obj", "cbox_test_1.obj"):
    with open(os.path.join(REF, "model/cbox", name + ".edge")) as fh:
        j = json.load(fh)
    edges = sorted(j["EdgeList"], key=lambda e: e["EdgeId"])
    faces = sorted(j["FaceList"], key=lambda f: f["FaceId"])
    out[name] = {
        "EdgeNum": j["EdgeNum"], "FaceNum": j["FaceNum"],
        "edges": [[e["OriVertex"], e["DestVertex"]] for e in edges],
        "faces": [[f["EdgeId0"], f["EdgeId1"], f["EdgeId2"]] for f in faces],
    }
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "tests/golden/cbox_edge_tables.json")
with open(dst, "w") as fh:
    json.dump(out, fh, separators=(",", ":"))
print("wrote", dst)
