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


@pytest.fixture(autouse=True)
def _default_dtype_guard():
    import torch
    prev = torch.get_default_dtype()
    yield
    torch.set_default_dtype(prev)
