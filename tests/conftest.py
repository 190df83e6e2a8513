import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class DirectOracle:
    """What the GPU parity tests compare with: the records of the REAL reference (oracle/_ref/libbvref.so, built where
    /root/reference exists and shipped to the GPU box) wherever that library is present, else the C restatement's.
    chi2 (not observable through the reference's API) and the decision margins always come from the restatement, which
    is bit-identical to the reference on every observable field (tests/test_oracle_cpu.py)."""

    def __init__(self, res, ref):
        self._res, self._ref = res, ref
        self.direct = ref is not None

    def run(self, slab, maf, n_threads=1):
        exp, gexp = self._res.run(slab, maf, n_threads=n_threads)
        if self._ref is not None:
            rexp, gexp = self._ref.run(slab, maf, n_threads=n_threads)
            rexp = rexp.copy()
            rexp["chi2"] = exp["chi2"]
            exp = rexp
        return exp, gexp

    def run_with_margins(self, slab, maf, n_threads=1):
        exp, gexp, margins = self._res.run_with_margins(slab, maf, n_threads=n_threads)
        if self._ref is not None:
            rexp, gexp = self._ref.run(slab, maf, n_threads=n_threads)
            rexp = rexp.copy()
            rexp["chi2"] = exp["chi2"]
            exp = rexp
        return exp, gexp, margins

    def __getattr__(self, name):
        return getattr(self._res, name)


@pytest.fixture(scope="session")
def restatement():
    import oracle
    oracle.build(with_ref=True)  # compiles liboracle.so; _ref only where /root/reference exists
    res = oracle.Restatement()
    return DirectOracle(res, oracle.Reference() if oracle.ref_available() else None)


@pytest.fixture(scope="session")
def reference():
    import oracle
    if not oracle.ref_available():
        oracle.build(with_ref=True)
    if not oracle.ref_available():
        pytest.skip("oracle/_ref/libbvref.so not built and /root/reference absent")
    return oracle.Reference()
