import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Read-only view of one tests/golden/*.npz fixture with '/'-separated keys."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __getitem__(self, k):
        return self._z[k]

    def keys(self):
        return list(self._z.keys())

    def group(self, prefix):
        prefix = prefix.rstrip("/") + "/"
        return {k[len(prefix):]: self._z[k] for k in self._z.keys() if k.startswith(prefix)}


_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture(scope="session")
def load_golden():
    return golden


def record_achieved(name, value):
    """Append an achieved error of a toleranced comparison to gpurun_out/parity_achieved.jsonl (scratch, merged back by
    gpurun): the tolerances in the tests are bounds, this is what the kernels actually reached on the box."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_achieved.jsonl"), "a") as f:
            f.write(json.dumps({"name": name, "value": float(value)}) + "\n")
    except OSError:
        pass
