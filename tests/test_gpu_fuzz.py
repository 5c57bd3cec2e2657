"""A few seeds of the two randomised parity checks (tools/fuzz_parity.py, tools/fuzz_render.py; run
on an MI355X with -m gpu): random triangle soups with adversarial rays through ptx_raycast /
ptx_vistest, and the Cornell box with random geometry, materials and lights through ptx_render in
both estimators, against the CPU oracle's plain loop over all faces."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_random_soups_and_adversarial_rays(root):
    import fuzz_parity
    from oracle import scene_io
    template = scene_io.load_conf(os.path.join(root, "config/cbox.conf"), root_dir=root, width=8, height=8, spp=1)
    sizes = set()
    for seed in (1, 2, 3, 5, 8, 13, 21, 34):
        n, bad, hit, clear, in_plane = fuzz_parity.one(seed, 2500, template)
        assert bad == 0, (seed, n, bad)
        assert 0.1 < hit < 0.95
        sizes.add(n <= 64)
    assert sizes == {True, False}          # both intersectors were exercised


def test_random_scenes_render_like_the_oracle(root, monkeypatch, capsys):
    import fuzz_render
    monkeypatch.setattr(sys, "argv", ["fuzz_render.py", "--seeds", "8", "--first", "100", "--size", "20"])
    assert fuzz_render.main() == 0
    assert "TOTAL failures: 0" in capsys.readouterr().out
