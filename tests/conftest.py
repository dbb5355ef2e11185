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


_PARITY = {}


def parity_record(key, **values):
    """Measured parity numbers of a GPU test run (max |dlogit|, fractions of identical rows, ...).  Written at session end to
    gpurun_out/parity_r6.json (merged back from the GPU box); the copy under profiles/ is the tracked record DESIGN.md quotes."""
    _PARITY.setdefault(key, {}).update({k: (float(v) if isinstance(v, (float, np.floating)) else int(v) if isinstance(v, (int, np.integer)) else v)
                                        for k, v in values.items()})


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_r6.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_PARITY)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def orc():
    from oracle import scp_oracle
    scp_oracle.lib()
    return scp_oracle
