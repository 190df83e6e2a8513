"""TEST INFRASTRUCTURE -- ctypes loaders for the two CPU checkers.

* ``liboracle.so``      plain-C restatement (oracle/refcpu.c)
* ``_ref/libbvref.so``  the real reference hot path (oracle/ref_driver.cpp + the
                        reference's own sources, compiled by oracle/Makefile)
* ``_ref/libbvcaller.so``  the reference's own per-position caller, batchfile rows in ->
                        the CVG / VCF bytes it writes (oracle/ref_caller_driver.cpp; loaded by
                        tests/ref_caller.py)

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may
import this package.  The product package ``basevar_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("BASEVAR_REFERENCE", "/root/reference")


class SiteResult(C.Structure):
    """Mirror of bv_site_result (include/basevar_amd.h)."""
    _fields_ = [
        ("depth", C.c_uint32 * 4), ("total_depth", C.c_uint32), ("status", C.c_uint32),
        ("cvg_sb", C.c_uint32 * 4), ("cvg_fs", C.c_double), ("cvg_sor", C.c_double),
        ("n_alt", C.c_uint8), ("alt", C.c_uint8 * 4), ("n_em", C.c_uint8), ("em_iters", C.c_uint16),
        ("af", C.c_double * 4), ("caf", C.c_double * 4), ("qual", C.c_double), ("chi2", C.c_double),
        ("qd", C.c_double), ("var_sb", C.c_uint32 * 4), ("var_fs", C.c_double), ("var_sor", C.c_double),
        ("mq_ranksum", C.c_double), ("rpr_ranksum", C.c_double), ("bq_ranksum", C.c_double),
    ]


SITE_DTYPE = np.dtype([
    ("depth", "<u4", 4), ("total_depth", "<u4"), ("status", "<u4"),
    ("cvg_sb", "<u4", 4), ("cvg_fs", "<f8"), ("cvg_sor", "<f8"),
    ("n_alt", "u1"), ("alt", "u1", 4), ("n_em", "u1"), ("em_iters", "<u2"),
    ("af", "<f8", 4), ("caf", "<f8", 4), ("qual", "<f8"), ("chi2", "<f8"), ("qd", "<f8"),
    ("var_sb", "<u4", 4), ("var_fs", "<f8"), ("var_sor", "<f8"),
    ("mq_ranksum", "<f8"), ("rpr_ranksum", "<f8"), ("bq_ranksum", "<f8"),
])
GROUP_DTYPE = np.dtype([("n_alt", "u1"), ("alt", "u1", 4), ("reserved", "u1", 3), ("total_depth", "<u4"),
                        ("reserved2", "<u4"), ("af", "<f8", 4)])
assert SITE_DTYPE.itemsize == 208 and C.sizeof(SiteResult) == 208
assert GROUP_DTYPE.itemsize == 48


def build(with_ref=True):
    """Compile liboracle.so and (when the reference sources are present) _ref/libbvref.so."""
    targets = ["liboracle.so"]
    if with_ref and os.path.exists(os.path.join(REFERENCE_ROOT, "src", "basetype.cpp")):
        targets.append("ref")
    subprocess.check_call(["make", "-s", "-C", HERE, "REF=" + REFERENCE_ROOT] + targets)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_RUN_ARGS = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
             C.c_uint32, C.c_uint32, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_int]


class _Checker:
    """Common slab-in / records-out call for both libraries."""
    _fn = None
    _has_err = False

    def run(self, slab, min_af, n_threads=1):
        """slab: dict with numpy planes base_strand, qual, mapq, rpr ([S][pitch]), ref_base [S],
        optional group_id [N] and n_groups.  Returns (site records, group records or None)."""
        bs = np.ascontiguousarray(slab["base_strand"], dtype=np.uint8)
        S, pitch = bs.shape
        N = int(slab.get("n_samples", pitch))
        q = np.ascontiguousarray(slab["qual"], dtype=np.uint8)
        mq = slab.get("mapq")
        rp = slab.get("rpr")
        mq = None if mq is None else np.ascontiguousarray(mq, dtype=np.uint8)
        rp = None if rp is None else np.ascontiguousarray(rp, dtype=np.uint16)
        ref = np.ascontiguousarray(slab["ref_base"], dtype=np.uint8)
        gid = slab.get("group_id")
        ng = int(slab.get("n_groups", 0)) if gid is not None else 0
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint8)
        out = np.zeros(S, dtype=SITE_DTYPE)
        gout = np.zeros((S, ng), dtype=GROUP_DTYPE) if ng else None
        args = [_ptr(bs), _ptr(q), _ptr(mq), _ptr(rp), _ptr(ref), _ptr(gid), ng, S, N, pitch,
                float(min_af), _ptr(out), _ptr(gout), int(n_threads)]
        if self._has_err:
            err = C.create_string_buffer(512)
            rc = self._fn(*args, err, 512)
            if rc != 0:
                raise RuntimeError(err.value.decode())
        else:
            rc = self._fn(*args)
            if rc != 0:
                raise RuntimeError("oracle_run failed")
        return out, gout


class Restatement(_Checker):
    """oracle/refcpu.c"""

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_run.restype = C.c_int
        L.oracle_run.argtypes = _RUN_ARGS
        self._fn = L.oracle_run
        L.oracle_run_ex.restype = C.c_int
        L.oracle_run_ex.argtypes = _RUN_ARGS + [C.c_void_p]
        L.oracle_chi2_test.restype = C.c_double
        L.oracle_chi2_test.argtypes = [C.c_double, C.c_double]
        L.oracle_norm_dist.restype = C.c_double
        L.oracle_norm_dist.argtypes = [C.c_double]
        L.oracle_fisher_exact_test.restype = C.c_double
        L.oracle_fisher_exact_test.argtypes = [C.c_int] * 4
        L.oracle_wilcoxon.restype = C.c_double
        L.oracle_wilcoxon.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_min_af.restype = C.c_double
        L.oracle_min_af.argtypes = [C.c_uint32, C.c_float]

    def run_with_margins(self, slab, min_af, n_threads=1):
        """As run(); also returns, per site, the smallest gap that decided a discrete choice in the
        LRT (argmin runner-up or distance to the threshold 24), including the site's group calls.
        Diagnostic for the parity tests: sites with a rounding-noise margin are order-dependent ties
        in the reference and are reported as ambiguous instead of being compared."""
        S = np.asarray(slab["base_strand"]).shape[0]
        margins = np.full(S, np.inf, dtype=np.float64)
        fn = self._fn
        ex = self.lib.oracle_run_ex
        self._fn = lambda *a: ex(*a, _ptr(margins))
        try:
            out, gout = self.run(slab, min_af, n_threads)
        finally:
            self._fn = fn
        return out, gout, margins

    def chi2_test(self, x, df=1.0):
        return self.lib.oracle_chi2_test(x, df)

    def norm_dist(self, x):
        return self.lib.oracle_norm_dist(x)

    def fisher(self, a, b, c, d):
        return self.lib.oracle_fisher_exact_test(a, b, c, d)

    def wilcoxon(self, s1, s2):
        a = np.ascontiguousarray(s1, dtype=np.float64)
        b = np.ascontiguousarray(s2, dtype=np.float64)
        return self.lib.oracle_wilcoxon(_ptr(a), len(a), _ptr(b), len(b))

    def min_af(self, n_samples, user_min_af=0.01):
        return self.lib.oracle_min_af(n_samples, user_min_af)


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libbvref.so"))


class Reference(_Checker):
    """oracle/_ref/libbvref.so: the real reference code behind oracle/ref_driver.cpp."""
    _has_err = True

    def run_timed(self, slab, min_af, n_threads=1):
        """As run(); also returns per-thread seconds spent inside the reference's path proper
        (the driver's slab -> BatchInfo conversion is excluded)."""
        secs = np.zeros(max(1, int(n_threads)), dtype=np.float64)
        fn = self._fn
        lib_fn = self.lib.bvref_run_timed
        self._fn = lambda *a: lib_fn(*a, _ptr(secs))
        try:
            out, gout = self.run(slab, min_af, n_threads)
        finally:
            self._fn = fn
        return out, gout, secs

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libbvref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (build it with `make -C oracle ref` where /root/reference exists)")
        L = self.lib = C.CDLL(path)
        L.bvref_run.restype = C.c_int
        L.bvref_run.argtypes = _RUN_ARGS + [C.c_char_p, C.c_size_t]
        self._fn = L.bvref_run
        L.bvref_run_timed.restype = C.c_int
        L.bvref_run_timed.argtypes = _RUN_ARGS + [C.c_char_p, C.c_size_t, C.c_void_p]
        L.bvref_chi2_test.restype = C.c_double
        L.bvref_chi2_test.argtypes = [C.c_double, C.c_double]
        L.bvref_norm_dist.restype = C.c_double
        L.bvref_norm_dist.argtypes = [C.c_double]
        L.bvref_fisher_exact_test.restype = C.c_double
        L.bvref_fisher_exact_test.argtypes = [C.c_int] * 4
        L.bvref_wilcoxon.restype = C.c_double
        L.bvref_wilcoxon.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.bvref_ranksum_int.restype = C.c_double
        L.bvref_ranksum_int.argtypes = [C.c_char, C.c_char_p, C.c_char_p, C.c_void_p, C.c_int]
        L.bvref_strand_bias.restype = C.c_int
        L.bvref_strand_bias.argtypes = [C.c_char, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]

    def chi2_test(self, x, df=1.0):
        return self.lib.bvref_chi2_test(x, df)

    def norm_dist(self, x):
        return self.lib.bvref_norm_dist(x)

    def fisher(self, a, b, c, d):
        return self.lib.bvref_fisher_exact_test(a, b, c, d)

    def wilcoxon(self, s1, s2):
        a = np.ascontiguousarray(s1, dtype=np.float64)
        b = np.ascontiguousarray(s2, dtype=np.float64)
        return self.lib.bvref_wilcoxon(_ptr(a), len(a), _ptr(b), len(b))
