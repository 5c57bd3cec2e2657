#!/usr/bin/env python3
"""Writes the golden vectors under tests/golden/ from the CPU oracle.

The reference ships no known-answer vectors for its tracer (its only pinning artefacts, the .edge
tables, are condensed by tools/extract_edge_fixture.py), so these files freeze the ORACLE's
outputs: they guard the restatement against accidental change and give the GPU tests inputs and
expected outputs that do not need the oracle to be rebuilt.

  tri_kat.npz          ray/triangle known answers: o,d,p1,p2,p3 -> accept, t, point
                       (random, on-edge, on-vertex, near-parallel, degenerate, behind-origin)
  samples.npz          hemisphere / light / BxDF tables keyed by explicit xi inputs
  cbox_first_hit.npz   first-hit face ids of the Cornell box at 64x64 and 256x256
  cbox_32_spp16.npz    32x32 x 16 spp pre-clamp image sums of the Cornell box (seed 13) and stats
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle, scene_io  # noqa: E402

G = os.path.join(ROOT, "tests/golden")
os.makedirs(G, exist_ok=True)


def tri_cases():
    rng = np.random.default_rng(7)
    o, d, p1, p2, p3 = [], [], [], [], []

    def add(oo, dd, a, b, c):
        o.append(oo), d.append(dd), p1.append(a), p2.append(b), p3.append(c)

    for _ in range(300):
        tri = rng.uniform(-2, 2, (3, 3)).astype(np.float32).astype(np.float64)
        oo = rng.uniform(-4, 4, 3)
        w = rng.dirichlet([1, 1, 1]) if rng.random() < 0.6 else rng.uniform(-0.5, 1.5, 3)
        tgt = w @ tri / w.sum()
        dd = tgt - oo
        dd /= np.linalg.norm(dd)
        if rng.random() < 0.15:
            dd = -dd  # behind the origin
        add(oo, dd, *tri)
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float64)
    for x, y in [(0.5, 0.0), (0.0, 0.5), (0.5, 0.5), (0, 0), (1, 0), (0, 1), (1e-11, 0.3),
                 (-1e-11, 0.3), (-1e-9, 0.3), (0.3, -5e-11), (0.5 + 4e-11, 0.5), (0.5 + 1e-9, 0.5),
                 (0.25, 0.25), (2.0, 2.0)]:
        add(np.array([x, y, 1.0]), np.array([0.0, 0.0, -1.0]), *tri)
        add(np.array([x + 0.3, y - 0.2, 1.0]), np.array([-0.3, 0.2, -1.0]) / np.linalg.norm([-0.3, 0.2, -1.0]), *tri)
    add(np.array([0.2, 0.2, 1.0]), np.array([1.0, 0, 0]), *tri)           # parallel
    add(np.array([0.2, 0.2, 1.0]), np.array([1.0, 0, -1e-12]), *tri)      # nearly parallel
    add(np.array([0.2, 0.2, 0.0]), np.array([0, 0, -1.0]), *tri)          # origin on the plane
    add(np.array([0.2, 0.2, -1e-11]), np.array([0, 0, -1.0]), *tri)       # just behind (t > -1e-10)
    add(np.array([0.2, 0.2, -1e-9]), np.array([0, 0, -1.0]), *tri)        # behind
    add(np.array([0.2, 0.2, 1.0]), np.array([0, 0, -1.0]), tri[0], tri[0], tri[1])  # degenerate
    add(np.array([0.2, 0.2, 1.0]), np.array([0, 0, 0.0]), *tri)           # zero direction
    o, d, p1, p2, p3 = map(np.array, (o, d, p1, p2, p3))
    acc = np.zeros(len(o), dtype=np.int8)
    t = np.zeros(len(o))
    pt = np.zeros((len(o), 3))
    for i in range(len(o)):
        ok, p = oracle.tri_raycast(o[i], d[i], p1[i], p2[i], p3[i])
        acc[i] = ok
        pt[i] = p
        t[i] = oracle.tri_raycast_t(o[i], d[i], p1[i], p2[i], p3[i])
    np.savez_compressed(os.path.join(G, "tri_kat.npz"), o=o, d=d, p1=p1, p2=p2, p3=p3, accept=acc, t=t, pt=pt)
    print("tri_kat:", len(o), "cases,", int(acc.sum()), "accepted")


def sample_tables():
    rng = np.random.default_rng(11)
    normals = [np.array(v, dtype=np.float64) for v in
               [(0, 1, 0), (0, -1, 0), (1, 0, 0), (0, 0, -1), (-0.999958, 0.008746, -0.002861)]]
    for _ in range(8):
        v = rng.normal(size=3)
        normals.append(v / np.linalg.norm(v))
    normals = [n / np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]) for n in normals]
    xi = rng.random((16, 2))
    xi[0] = (0, 0)
    xi[1] = (1 - 2 ** -53, 0.5)
    hemi = np.array([[oracle.sample_hemisphere(n, a, b) for a, b in xi] for n in normals])

    sq = np.zeros(16)
    sq[1:4] = 10
    sq[4:16] = np.array([[2.13, 5.48, 2.27], [3.43, 5.48, 2.27], [3.43, 5.48, 3.32], [2.13, 5.48, 3.32]]).ravel()
    sp = np.zeros(16)
    sp[0] = 1
    sp[1:4] = (20, 10, 20)
    sp[4:8] = (0.95, 3.6, -2.0, 0.35)
    refs = np.concatenate([rng.uniform(0, 5.5, (12, 3)), [[2.8, 5.6, 2.8], [1.0, 3.5, -2.1], [0.95, 3.6, -2.0]]])
    sq_out = np.array([[np.concatenate([[pdf], ro, rd]) for (pdf, ro, rd) in
                        [oracle.square_light_sample(sq, r, a, b)]][0] for r in refs for a, b in xi[:4]])
    sp_out = np.array([[np.concatenate([[pdf], ro, rd]) for (pdf, ro, rd) in
                        [oracle.sphere_light_sample(sp, r, a, b, 0.5 * (a + b))]][0] for r in refs for a, b in xi[:4]])

    mats = np.zeros((4, 12))
    mats[0, 3:6] = (0.85, 0.75, 0.65); mats[0, 9] = 1; mats[0, 10] = 1                 # diffuse
    mats[1, 3:6] = (0.77, 0.79, 0.73); mats[1, 6:9] = (0.97, 0.99, 0.93); mats[1, 9] = 1; mats[1, 10] = 98  # glossy
    mats[2, 9] = 1.5; mats[2, 10] = 1                                                  # glass
    mats[3, 3:6] = (0.1, 0.2, 0.3); mats[3, 6:9] = (0.5, 0.5, 0.5); mats[3, 9] = 1; mats[3, 10] = 7.0
    ev, sm = [], []
    for m in mats:
        for n in normals[:6]:
            for k in range(6):
                wi = rng.normal(size=3); wi /= np.sqrt(wi @ wi)
                wo = rng.normal(size=3); wo /= np.sqrt(wo @ wo)
                ev.append(np.concatenate([m, n, wi, wo, oracle.bxdf_eval(m, wi, wo, n)]))
                x = rng.random(2)
                w2, coef = oracle.bxdf_sample(m, n, wo, x)
                sm.append(np.concatenate([m, n, wo, x, w2, coef]))
    refr = []
    for _ in range(40):
        n = rng.normal(size=3); n /= np.sqrt(n @ n)
        i = rng.normal(size=3); i /= np.sqrt(i @ i)
        eta = [1 / 1.5, 1.5, 1.0][rng.integers(3)]
        refr.append(np.concatenate([n, i, [eta], oracle.refract(n, i, eta), oracle.reflect(n, i)]))
    np.savez_compressed(os.path.join(G, "samples.npz"), normals=np.array(normals), xi=xi, hemi=hemi,
                        square_light=sq, sphere_light=sp, refs=refs, square_out=sq_out, sphere_out=sp_out,
                        bxdf_eval=np.array(ev), bxdf_sample=np.array(sm), refract=np.array(refr))
    print("samples: hemi", hemi.shape, "eval", len(ev), "sample", len(sm))


def cbox_vectors():
    conf = os.path.join(ROOT, "config/cbox.conf")
    out = {}
    for w in (64, 256):
        sc = scene_io.load_conf(conf, root_dir=ROOT, width=w, height=w, spp=1)
        out["ids_%d" % w] = oracle.Oracle(sc).first_hit().astype(np.int8)
    np.savez_compressed(os.path.join(G, "cbox_first_hit.npz"), **out)
    sc = scene_io.load_conf(conf, root_dir=ROOT, width=32, height=32, spp=16)
    img, st = oracle.Oracle(sc, seed=13).render(stats=True)
    np.savez_compressed(os.path.join(G, "cbox_32_spp16.npz"), img=img,
                        stats=np.array([st["paths"], st["segments"], st["shadow_rays"]], dtype=np.int64))
    print("cbox: first-hit maps 64/256, image 32x32x16spp, stats", st)


if __name__ == "__main__":
    tri_cases()
    sample_tables()
    cbox_vectors()
