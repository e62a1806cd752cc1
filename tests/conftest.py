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
    # the CPU oracle legs run on torch's CPU pool: one thread per visible core under a cgroup CPU quota of a sixteenth of them is throttled
    # in every op (point2cyl_amd/hostmem.py) - size the pool to the quota for the whole session
    from point2cyl_amd import hostmem
    hostmem.fit_threads_to_quota(reserve=0)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture
def golden():
    return load_golden


def same_up_to_sign(a, b):
    """|a.b| for unit vectors along the last axis."""
    return np.abs((a * b).sum(-1))
