"""CPU tests (no GPU): pin the oracle.

1. oracle/refcpu.c (the restatement) == committed golden vectors, which were produced by the
   REAL reference compiled from /root/reference (tests/golden/make_golden.py).
2. restatement == real reference on fresh seeded slabs, when oracle/_ref is available
   (always in the build container; on the GPU box too, since the built .so travels).
3. known answers for the inputs of the reference's own tests/io/test_algorithm.cpp:13-31.
4. independent cross-check of the special functions against scipy.
"""
import glob
import json
import os

import numpy as np
import pytest

import oracle
from basevar_amd.synth import make_slab

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
EXACT_SKIP = ("chi2", "em_iters", "n_em")  # not observable through the reference's public API


def load_fixture(path):
    d = np.load(path)
    slab = {k: d[k] for k in ("base_strand", "qual", "mapq", "rpr", "ref_base")}
    slab["n_samples"] = int(d["n_samples"])
    slab["n_groups"] = int(d["n_groups"])
    if "group_id" in d.files:
        slab["group_id"] = d["group_id"]
    exp = d["expected_sites"].view(oracle.SITE_DTYPE).reshape(-1)
    gexp = None
    if "expected_groups" in d.files:
        gexp = d["expected_groups"].view(oracle.GROUP_DTYPE).reshape(len(exp), -1)
    return slab, float(d["min_af"]), exp, gexp


def assert_bit_equal(a, b, skip=()):
    for f in a.dtype.names:
        if f in skip or f.startswith("reserved"):
            continue
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            assert np.array_equal(x, y, equal_nan=True), "field %s differs" % f
        else:
            assert np.array_equal(x, y), "field %s differs" % f


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))), ids=os.path.basename)
def test_restatement_matches_golden(restatement, path):
    slab, maf, exp, gexp = load_fixture(path)
    got, ggot = restatement.run(slab, maf)
    assert_bit_equal(got, exp, skip=EXACT_SKIP)
    if gexp is not None:
        assert_bit_equal(ggot, gexp)


@pytest.mark.parametrize("seed,n,cov,groups", [(101, 1500, 0.3, 2), (102, 6000, 0.08, 0), (103, 777, 0.9, 3)])
def test_restatement_matches_reference_fresh(restatement, reference, seed, n, cov, groups):
    slab = make_slab(120, n, seed=seed, coverage=cov, n_groups=groups, ref_n_frac=0.05)
    maf = restatement.min_af(n)
    a, ga = restatement.run(slab, maf)
    b, gb = reference.run(slab, maf)
    assert_bit_equal(a, b, skip=EXACT_SKIP)
    if groups:
        assert_bit_equal(ga, gb)
    assert ((a["status"] & 2) != 0).sum() > 5  # the slab exercises the variant branch


def test_restatement_threads_agree(restatement):
    slab = make_slab(64, 2000, seed=7, coverage=0.2, n_groups=2)
    maf = restatement.min_af(2000)
    a, ga = restatement.run(slab, maf, n_threads=1)
    b, gb = restatement.run(slab, maf, n_threads=4)
    assert_bit_equal(a, b)
    assert_bit_equal(ga, gb)


def test_known_answers(restatement):
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    for x, df, v in ka["chi2_test"]:
        got = restatement.chi2_test(x, df)
        assert (np.isnan(got) and (v is None or v != v)) or got == v
    for x, v in ka["norm_dist"]:
        assert restatement.norm_dist(x) == v
    for t, v in ka["fisher_exact_test"]:
        assert restatement.fisher(*t) == v
    for s1, s2, v in ka["wilcoxon_ranksum_test"]:
        assert restatement.wilcoxon(s1, s2) == v
    # the values quoted in SURVEY.md section 4
    assert restatement.wilcoxon([1, 5, 3, 10, 3, 3, 4, 5], [6, 7, 2, 2, 8, 9, 10]) == pytest.approx(0.27158867424337468, rel=1e-15)
    assert restatement.fisher(345, 455, 260, 345) == pytest.approx(0.95667786399050136, rel=1e-15)
    assert restatement.chi2_test(24.0) == pytest.approx(9.633570086430948e-07, rel=1e-15)
    assert restatement.norm_dist(1.96) == pytest.approx(0.024997895148220445, rel=1e-15)


def test_restated_kfunc_on_a_few_hand_picked_arguments(restatement):
    """Independent check that the restated kfunc algorithms compute what they claim."""
    from scipy import stats
    for x in (0.5, 3.84, 10.0, 24.0, 100.0, 700.0):
        assert restatement.chi2_test(x) == pytest.approx(stats.chi2.sf(x, 1), rel=1e-9)
    for x in (0.0, 0.5, 1.96, 5.0):
        assert restatement.norm_dist(x) == pytest.approx(stats.norm.sf(x), rel=1e-6)  # AS66: ~1e-7 accurate
    for t in [(8, 4, 4, 9), (10, 5, 4, 9), (1200, 1300, 900, 700), (3, 0, 0, 3)]:
        p = stats.fisher_exact([[t[0], t[1]], [t[2], t[3]]])[1]
        assert restatement.fisher(*t) == pytest.approx(p, rel=1e-6)


def test_min_af_is_float_rounded(restatement):
    assert restatement.min_af(100000) == 0.0010000000474974513  # SURVEY trap #2
    assert restatement.min_af(100) == float(np.float32(0.01))
    assert restatement.min_af(1000000) == float(np.float32(100.0) / np.float32(1000000.0))


def test_special_functions_against_scipy(restatement):
    """Third, independent pin of the special functions the path takes from htslib/kfunc.c (kf_gammaq, kf_erfc,
    kt_fisher_exact) and of the Wilcoxon z-test: the restatement (and, where the build container has it, the compiled
    reference behind known_answers.json) against scipy's implementations of the same published functions."""
    import json
    from scipy import special, stats
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    for x, df, ref_val in ka["chi2_test"]:
        if x < 0:
            continue
        want = special.gammaincc(df / 2.0, x / 2.0)  # chi-square survival function, algorithm.h:44-46
        assert ref_val == pytest.approx(want, rel=1e-9, abs=1e-300)
        assert restatement.chi2_test(x, df) == pytest.approx(want, rel=1e-9, abs=1e-300)
    for x, ref_val in ka["norm_dist"]:
        want = 0.5 * special.erfc(x / np.sqrt(2.0))  # algorithm.h:48-50
        assert ref_val == pytest.approx(want, rel=1e-6, abs=1e-300)  # AS66 is a 1e-7-class approximation
        assert restatement.norm_dist(x) == pytest.approx(want, rel=1e-6, abs=1e-300)
    tables = [t for t, _ in ka["fisher_exact_test"]] + [[3, 1, 1, 3], [12, 0, 0, 9], [100, 200, 150, 90], [7, 5, 0, 12]]
    refv = {tuple(t): v for t, v in ka["fisher_exact_test"]}
    for t in tables:
        want = stats.fisher_exact([[t[0], t[1]], [t[2], t[3]]], alternative="two-sided")[1]
        got = restatement.fisher(*t)
        assert got == pytest.approx(want, rel=1e-7, abs=1e-300), t
        if tuple(t) in refv:
            assert refv[tuple(t)] == pytest.approx(want, rel=1e-7, abs=1e-300), t
    rng = np.random.default_rng(1)
    for _ in range(20):
        a = rng.integers(0, 60, int(rng.integers(3, 40))).astype(float)
        b = rng.integers(10, 70, int(rng.integers(3, 40))).astype(float)
        # algorithm.h:76-136: normal approximation without tie or continuity correction, two-sided
        n1, n2 = len(a), len(b)
        ranks = stats.rankdata(-np.concatenate([a, b]))  # descending, average ranks
        z = (ranks[:n1].sum() - n1 * (n1 + n2 + 1) / 2.0) / np.sqrt(n1 * n2 * (n1 + n2 + 1) / 12.0)
        want = 2 * 0.5 * special.erfc(abs(z) / np.sqrt(2.0))
        assert restatement.wilcoxon(list(a), list(b)) == pytest.approx(want, rel=1e-6, abs=1e-300)
