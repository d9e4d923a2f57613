import os
import sys

import pytest

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
