import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The oracle (gcc) is cheap to build, so make sure it exists; the HIP library is built by
    __graft_entry__.build() / `make -C pathtracer_amd/csrc` beforehand and only checked here."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "scenes", "cbox", "cbox.obj")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "--only", "cbox,mis"])
    yield


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def room_scene():
    """The ~250 k-triangle room stand-in is generated on demand (25 MB, not committed)."""
    p = os.path.join(ROOT, "scenes", "room", "room.obj")
    if not os.path.exists(p):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_scenes.py"), "--only", "room"])
    return p


def pytest_terminal_summary(terminalreporter):
    """f64 image comparisons of the GPU parity tests: how many pixels differed beyond the bar (each
    sample of each was re-traced by the oracle and equals a named tie alternative of its path, or the
    test failed)."""
    mod = sys.modules.get("test_gpu_parity")
    rows = getattr(mod, "OUTLIERS", None) if mod else None
    if rows:
        total = sum(r[1] for r in rows)
        terminalreporter.write_line("f64 parity: %d image comparisons, %d pixels beyond 1e-9 in all (every one matched to a named tie "
                                    "alternative, or -- %d of them -- to the amplification the oracle measures on the path):"
                                    % (len(rows), total, sum(r[4] for r in rows)))
        for name, bad, n, proven, amplified in rows:
            if bad:
                terminalreporter.write_line("  %s: %d of %d pixels, %d proven ties, %d amplified roundings" % (name, bad, n, proven, amplified))
