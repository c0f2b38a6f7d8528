import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "index-tts-ipex_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLD, name + ".npz")))

    return load


@pytest.fixture(scope="session")
def accuracy():
    """Measured accuracy numbers of the GPU parity tests (bf16 engine vs the reference fixtures ...), collected in
    one dict and written to gpurun_out/r04_accuracy.json when the session ends (copied to profiles/ and committed)."""
    import json

    rec = {}
    yield rec
    if rec:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r04_accuracy.json"), "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
