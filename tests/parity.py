"""Record-by-record comparison of engine output with an oracle's (test helper).

Bar (BASELINE.json north_star): integer depth / allele / strand counts bit-exact;
AF / QUAL / LRT-derived floats within 1e-6 relative.  FS and the rank-sum phred values
can legitimately be ~1e-15 (p == 1 up to rounding), so they also pass on |diff| <= 1e-9.
"""
import numpy as np

RTOL = 1e-6
ATOL_PHRED = 1e-9

INT_FIELDS = ["depth", "total_depth", "cvg_sb", "n_alt", "alt", "var_sb"]
REL_FIELDS = ["af", "caf", "qual", "qd", "cvg_sor", "var_sor"]
PHRED_FIELDS = ["cvg_fs", "var_fs", "mq_ranksum", "rpr_ranksum", "bq_ranksum"]
STATUS_MASK = 0x1 | 0x2 | 0x4 | 0x8  # covered, variant, bad-qual, zero-freq


def _close(x, y, rtol, atol):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    both_nan = np.isnan(x) & np.isnan(y)
    same_inf = np.isinf(x) & np.isinf(y) & (np.sign(x) == np.sign(y))
    with np.errstate(invalid="ignore"):
        ok = np.abs(x - y) <= np.maximum(atol, rtol * np.abs(y))
    return ok | both_nan | same_inf


TIE_EPS = 1e-9


def ambiguous_sites(exp, margins=None):
    """Sites whose discrete outcome hinges on a floating-point tie.  `margins` comes from
    oracle.Restatement.run_with_margins(): the smallest gap that decided an argmin or the
    `chi2 < 24` test (basetype.cpp:157-161) anywhere in the site's LRT (and its group calls).  When
    two allele subsets have mathematically equal likelihood (e.g. two bases seen once each with the
    same phred), the reference's choice depends on the ORDER in which it adds the per-sample
    log-likelihoods; the engine works on order-free histograms.  Such sites are reported, never
    silently dropped: callers assert that they are rare."""
    chi = exp["chi2"]
    with np.errstate(invalid="ignore"):
        amb = np.abs(chi - 24.0) <= 24.0 * TIE_EPS
    if margins is not None:
        scale = np.maximum(1.0, np.abs(np.nan_to_num(chi, nan=0.0)))
        amb = amb | (np.asarray(margins) <= TIE_EPS * scale)
    return amb


def compare_sites(got, exp, check_ranks=True, check_chi2=True):
    """Returns {field: array of mismatching site indices} (empty dict == parity)."""
    bad = {}
    n = len(exp)
    assert len(got) == n

    def rows(mask):
        mask = np.asarray(mask)
        if mask.ndim > 1:
            mask = mask.reshape(n, -1).any(axis=1)
        return np.nonzero(mask)[0]

    for f in INT_FIELDS:
        m = rows(got[f] != exp[f])
        if m.size:
            bad[f] = m
    m = rows((got["status"] & STATUS_MASK) != (exp["status"] & STATUS_MASK))
    if m.size:
        bad["status"] = m
    for f in REL_FIELDS:
        m = rows(~_close(got[f], exp[f], RTOL, 0.0))
        if m.size:
            bad[f] = m
    for f in PHRED_FIELDS:
        if not check_ranks and f in ("mq_ranksum", "rpr_ranksum"):
            continue
        m = rows(~_close(got[f], exp[f], RTOL, ATOL_PHRED))
        if m.size:
            bad[f] = m
    if check_chi2:
        m = rows(~_close(got["chi2"], exp["chi2"], RTOL, 1e-7))
        if m.size:
            bad["chi2"] = m
    return bad


def compare_groups(got, exp, variant_mask):
    bad = {}
    if exp is None:
        return bad
    v = np.nonzero(variant_mask)[0]
    if v.size == 0:
        return bad
    g, e = got[v], exp[v]
    for f in ["n_alt", "alt", "total_depth"]:
        m = (g[f] != e[f]).reshape(len(v), -1).any(axis=1)
        if m.any():
            bad["group." + f] = v[m]
    m = (~_close(g["af"], e["af"], RTOL, 0.0)).reshape(len(v), -1).any(axis=1)
    if m.any():
        bad["group.af"] = v[m]
    return bad


def describe(bad, got, exp, limit=3):
    lines = []
    for f, idx in bad.items():
        lines.append("%s: %d site(s) differ, first %s" % (f, len(idx), idx[:limit].tolist()))
        fld = f.split(".")[-1]
        if not f.startswith("group."):
            for i in idx[:limit]:
                lines.append("   site %d got=%r exp=%r  (exp n_alt=%d chi2=%r depth=%r)" % (
                    i, got[fld][i].tolist() if hasattr(got[fld][i], "tolist") else got[fld][i],
                    exp[fld][i].tolist() if hasattr(exp[fld][i], "tolist") else exp[fld][i],
                    exp["n_alt"][i], float(exp["chi2"][i]), exp["depth"][i].tolist()))
    return "\n".join(lines)
