import os

# see pnode_amd/__init__.py: hipGraph replays of PyTorch reductions need this on ROCm 7.2
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracles run on the host.  On the GPU box (256 hardware threads) PyTorch's default intra-op pool of 128 threads makes
    # their small dense operations 15x SLOWER than 16 threads do (measured, tools/time_c5_test_pieces.py: the exact-Newton
    # ARKIMEX oracle of the C5 shard test 98.9 s against 6.4 s) -- the whole suite's time was mostly that.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the product library is current (a no-op when the stored source hash matches; hipcc cross-compiles
    without a GPU) and the oracle's C restatement is compiled (gcc, seconds)."""
    import __graft_entry__ as ge
    ge.build_library()
    from oracle import ts_oracle
    ts_oracle.build()


@pytest.fixture(autouse=True)
def _clean_options():
    """Every test starts with an empty options database."""
    from pnode_amd import options
    options.clear()
    yield
    options.clear()


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this container")
    return torch.device("cuda:0")
