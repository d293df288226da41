import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("MSHGNN_POISON_WS", "1")      # engine.workspace(): fresh workspaces are filled with 0xFF (NaNs, all-ones relu bytes) -- a kernel that reads what
                                                     # no launch wrote then fails its parity test instead of finding an earlier engine's values in recycled memory


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the fp64 oracle is small-matrix torch code: on a many-core host (the GPU box shows 256 cpus, 16 of them this job's share) the default of one
    # thread per visible core makes it SLOWER (bench.py's cpu_baseline scan: best at 16) -- cap it for every oracle evaluation of the suite
    import torch
    torch.set_num_threads(max(1, min(torch.get_num_threads(), 16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16)))


@pytest.fixture(autouse=True)
def _default_dtype_guard():
    import torch
    prev = torch.get_default_dtype()
    yield
    torch.set_default_dtype(prev)
