import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def engine():
    """The HIP engine.  Skips only when no GPU is present; a GPU box with a
    missing/broken libfss_amd.so must fail, not skip."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    import fss_amd
    return fss_amd.default_engine()


@pytest.fixture(scope="session")
def engine_plain(engine):
    """A second context with the same keys whose one-key point evaluations always walk from the root
    (FSS_AMD_POINTS_PLAIN): the reference the shared-prefix plan is compared with."""
    import fss_amd
    return fss_amd.Engine(engine.device.index).set_points_plan("plain")


class SplitMix64:
    """Deterministic synthetic-input stream."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def i4(self, clear_lsb=True):
        a, b = self.next(), self.next()
        v = np.array([a & 0xFFFFFFFF, a >> 32, b & 0xFFFFFFFF, b >> 32], dtype=np.uint32)
        if clear_lsb:
            v[3] &= 0xFFFFFFFE
        return v.view(np.int32)

    def below(self, bits):
        return self.next() & ((1 << bits) - 1) if bits < 64 else self.next()


@pytest.fixture
def rng():
    return SplitMix64(0x5EED0001)
