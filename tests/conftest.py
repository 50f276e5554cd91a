import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden():
    import torch
    return torch.load(os.path.join(GOLDEN, "golden.pt"), weights_only=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import bd_oracle
    bd_oracle.lib()
    return bd_oracle


@pytest.fixture(autouse=True)
def _seed_unseeded_draws(request):
    """Every test starts from its OWN global RNG state (a hash of its node id): draws that do not pass a generator are reproducible and do not
    depend on which tests ran before (a 1-in-40 tolerance miss in a residual check surfaced only under one test ordering -- round 5)."""
    import zlib
    import torch
    torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7fffffff)
    yield


try:        # property tests draw the SAME examples on every run (here, on the GPU box, in the driver's round-end run)
    from hypothesis import settings as _hyp_settings
    _hyp_settings.register_profile("reproducible", derandomize=True, deadline=None, database=None)
    _hyp_settings.load_profile("reproducible")
except ImportError:
    pass
