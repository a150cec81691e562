import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    # torch ships its own HIP runtime: when a test uses both (device tensors handed to the C ABI, as bench.py does),
    # torch's copy has to be the one that is loaded first, or torch finds no GPU afterwards
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(g.PKG_DIR, "lib", "libparasail_amd.so")):
        g.build()
    return g.load_pkg()


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as fh:
        return json.load(fh)["cases"]
