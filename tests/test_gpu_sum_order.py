"""The library built with the other plausible orders of the short sums (make SUM=1 / SUM=2 in
pathtracer_amd/csrc: libptx_amd_sum1.so, libptx_amd_sum2.so) against the CPU restatement built
with the same order: whichever order Eigen uses in the reference, the GPU path can follow it and
stays bit-exact with the restatement.  Each variant runs in a process of its own (one library per
process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECK = r"""
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import pathtracer_amd as pt
from oracle import oracle, scene_io
v = %(variant)d
assert ("sum order %%d" %% v).encode() in pt.lib().ptx_version()
oracle.use_sum_order(v)
for conf, w, h, spp in (("config/cbox.conf", 256, 256, 2), ("config/mis.conf", 192, 128, 2)):
    s = pt.Scene.load_conf(os.path.join(%(root)r, conf), root_dir=%(root)r, width=w, height=h, spp=spp)
    ref = scene_io.load_conf(os.path.join(%(root)r, conf), root_dir=%(root)r, width=w, height=h, spp=spp)
    c = pt.Context(0); c.build_accel(s)
    O = oracle.Oracle(ref, seed=13)
    o, d = c.generate_rays(); oo, od = O.generate_rays()
    assert np.array_equal(o, oo) and np.array_equal(d, od), "camera rays"
    ids, pts = c.first_hit(with_points=True); oids, opts = O.first_hit(with_points=True)
    assert np.array_equal(ids, oids) and np.array_equal(pts, opts, equal_nan=True), "first hits"
    img, st = c.render(mode="f64"); oimg = O.render()
    bad = int((np.abs(img - oimg).max(axis=-1) > 1e-9).sum())
    assert bad <= 4, "image: %%d pixels" %% bad
print("ok")
"""


@pytest.mark.parametrize("variant", [1, 2])
def test_library_follows_the_sum_order_of_the_restatement(variant):
    lib = os.path.join(ROOT, "pathtracer_amd", "libptx_amd_sum%d.so" % variant)
    if not os.path.exists(lib):
        pytest.skip("libptx_amd_sum%d.so not built (make -C pathtracer_amd/csrc SUM=%d)" % (variant, variant))
    env = dict(os.environ, PTX_LIB_OVERRIDE=lib)
    p = subprocess.run([sys.executable, "-c", CHECK % dict(root=ROOT, variant=variant)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-3000:]
