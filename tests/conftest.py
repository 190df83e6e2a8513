import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def restatement():
    import oracle
    oracle.build(with_ref=True)  # compiles liboracle.so; _ref only where /root/reference exists
    return oracle.Restatement()


@pytest.fixture(scope="session")
def reference():
    import oracle
    if not oracle.ref_available():
        oracle.build(with_ref=True)
    if not oracle.ref_available():
        pytest.skip("oracle/_ref/libbvref.so not built and /root/reference absent")
    return oracle.Reference()
