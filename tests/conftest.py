import os
import sys

import pytest

# PyTorch-ROCm bundles its own copy of the HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  Whichever copy a
# process loads FIRST is the one that gets the GPU: with torch imported first, libclownresampler_amd.so binds to torch's copy
# and both work; the other way round torch finds "No HIP GPUs".  Tests that use both (streams, graphs, device tensors) must
# not depend on which test file happened to be collected first.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - the library itself does not need it
    pass

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """The oracle is C: make sure it is compiled (and the in-place reference build, where /root/reference exists)."""
    import _checkers
    _checkers.build_checkers()
