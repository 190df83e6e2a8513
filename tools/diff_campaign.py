#!/usr/bin/env python3
"""Randomised differential campaign on a GPU box: engine (C ABI) vs the oracle on many seeded
slabs of varied shape/distribution.  Prints one summary line per slab and a total; exits 1 on
any non-ambiguous mismatch.  Not a pytest (minutes, not seconds)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import basevar_amd  # noqa: E402
import oracle  # noqa: E402
from basevar_amd.synth import make_slab, tag_ranks  # noqa: E402
from parity import ambiguous_sites, compare_groups, compare_sites, describe  # noqa: E402


class _Got:
    pass


def lrt_chained(eng, slab, rng, layout=0):
    """The slab cut into 2-5 row ranges, uploaded, and submitted as ONE chained launch (CAMPAIGN_CHAIN=1; pop-groups too)."""
    import torch
    dev = torch.device("cuda", 0)
    S, n, pitch = slab["n_sites"], slab["n_samples"], slab["pitch"]
    t = {k: torch.from_numpy(np.ascontiguousarray(slab[k])).to(dev) for k in ("base_strand", "qual", "ref_base", "mapq")}
    t["rpr"] = torch.from_numpy(np.ascontiguousarray(slab["rpr"]).view(np.int16)).to(dev)
    rec = basevar_amd.SITE_DTYPE.itemsize
    out = torch.zeros(S * rec, dtype=torch.uint8, device=dev)
    G = int(slab.get("n_groups", 0) or 0)
    grec = basevar_amd.GROUP_DTYPE.itemsize
    gid = torch.from_numpy(np.ascontiguousarray(slab["group_id"])).to(dev) if G else None
    gout = torch.zeros(max(1, S * G * grec), dtype=torch.uint8, device=dev)
    k = int(rng.integers(2, 6))
    cuts = sorted(set([0, S] + [int(c) for c in rng.integers(1, max(2, S), size=k - 1)]))
    segs = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        segs.append((hi - lo, t["base_strand"][lo].data_ptr(), t["qual"][lo].data_ptr(), t["ref_base"][lo:].data_ptr(),
                     out.data_ptr() + lo * rec, t["mapq"][lo].data_ptr(), t["rpr"][lo].data_ptr()))
    torch.cuda.synchronize()
    eng.submit_many_ptrs(n, pitch, segs, group_id=gid.data_ptr() if G else 0, n_groups=G,
                         gouts=[gout.data_ptr() + lo * G * grec for lo in cuts[:-1]] if G else None, layout=layout)
    eng.wait()
    g = _Got()
    g.sites = out.cpu().numpy().view(basevar_amd.SITE_DTYPE)
    g.groups = gout.cpu().numpy()[:S * G * grec].view(basevar_amd.GROUP_DTYPE).reshape(S, G) if G else None
    g.n_variant = int(((g.sites["status"] & 2) != 0).sum())
    return g


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    use_ref = oracle.ref_available() and os.environ.get("CAMPAIGN_ORACLE", "ref") == "ref"
    chk = oracle.Reference() if use_ref else oracle.Restatement()
    res = oracle.Restatement()
    rng = np.random.default_rng(int(os.environ.get("CAMPAIGN_SEED", "1")))
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    tot_sites = tot_var = tot_amb = tot_bad = tot_exc_site = tot_exc_group = 0
    t0 = time.time()
    for it in range(rounds):
        if os.environ.get("CAMPAIGN_FUSED") == "1":  # the row lengths of csrc/bv_pass1_fused.hip (4,097 .. 49,152 samples), ragged ones too
            n = int(rng.choice([4097, 5000, 6145, 8191, 10000, 12289, 16400, 20003, 30000, 40000, 49151, 49152]))
        elif os.environ.get("CAMPAIGN_SHALLOW") == "1":  # where order-dependent ties live: rows of a few covered samples
            n = int(rng.choice([8, 16, 37, 64, 120, 300, 1000]))
        elif os.environ.get("CAMPAIGN_TILES") == "1":
            n = int(rng.choice([8, 37, 64, 300, 2500, 10000, 40000]))
        else:
            n = int(rng.choice([37, 300, 2500, 10000, 40000, 49152, 49153, 60000, 120000, 300000, 1000000]))
        sites = int(max(16, min(4096, 6_000_000 // n)))
        cov = float(rng.choice([0.02, 0.08, 0.3, 0.9]))
        qm = float(rng.choice([10.0, 25.0, 33.0]))
        classes = []
        for _ in range(8):
            a = float(rng.choice([0, 0, 0.0005, 0.002, 0.01, 0.05, 0.2, 0.5, 0.95, 1.0]))
            b = float(rng.choice([0, 0, 0, 0.01, 0.1, 0.3]))
            classes.append((a, min(b, 1.0 - a)))
        ng = int(rng.choice([0, 0, 2, 5, 32, 64 if n <= 60000 else 2]))  # (64: two rounds of 32 groups, round 5)
        if os.environ.get("CAMPAIGN_TILES") == "1" and (int(os.environ.get("CAMPAIGN_FLAGS", "0"), 0) & 8):
            ng = min(ng, 32)  # (the per-site-tally realisation holds one histogram per group in the site's state: <= 32 groups)
        slab = make_slab(sites, n, seed=int(rng.integers(1 << 30)), coverage=cov, qual_mean=qm, qual_sd=9.0,
                         qual_min=1, qual_max=60, n_groups=ng, class_af=classes, ref_n_frac=0.03)
        if os.environ.get("CAMPAIGN_NORANKS") == "1":  # no mapq / rank planes: pass 1 alone (the fused kernel without its pass-2 rows)
            slab.pop("mapq"); slab.pop("rpr")
        maf = res.min_af(n, float(rng.choice([0.01, 0.001])))
        eng = basevar_amd.BaseTypeEngine(sites, maf, flags=int(os.environ.get("CAMPAIGN_FLAGS", "0"), 0))
        # round 6: every other slab reaches the engine in the tagged rank layout (BV_SLAB_RPR_TAGGED: the producer's choice -- ranks
        # here are <= 100); the oracle always gets the plain ranks
        fed = tag_ranks(slab) if ("rpr" in slab and rng.random() < 0.5) else slab
        if os.environ.get("CAMPAIGN_CHAIN") == "1":
            got = lrt_chained(eng, fed, rng, layout=int(fed.get("layout", 0)))  # the same rows as 2-5 slabs through bv_engine_submit_many
        elif os.environ.get("CAMPAIGN_TILES") == "1":
            # the sample axis in tiles of a random width (CAMPAIGN_FLAGS=8: the per-site-tally realisation; ranks up to 100: no overflow);
            # round 6: dense tiles, packed tiles (bv_engine_tiles_add_sparse) or a mix of the two
            got = eng.lrt_tiles(fed, int(rng.choice([7, 64, 200, 1000, max(16, n // 3)])), packed=[False, True, 3][int(rng.integers(0, 3))])
        else:
            got = eng.lrt(fed)
        eng.close()
        exp, gexp = chk.run(slab, maf, n_threads=threads)
        # decision margins always come from the restatement (bit-identical to the reference)
        exp_r, _, margins = res.run_with_margins(slab, maf, n_threads=threads)
        amb = ambiguous_sites(exp_r, margins)
        bad = compare_sites(got.sites, exp, check_chi2=not use_ref, check_ranks=os.environ.get("CAMPAIGN_NORANKS") != "1")
        bad.update(compare_groups(got.groups, gexp, (exp["status"] & 2) != 0))
        excused = set()
        exc_site, exc_group = set(), set()
        for f, idx in bad.items():
            excused.update(idx[amb[idx]].tolist())
            (exc_group if f.startswith("group.") else exc_site).update(idx[amb[idx]].tolist())
        for i in sorted(excused):
            gd = "" if gexp is None else " group depths %s" % gexp[i]["total_depth"].tolist()
            if os.environ.get("CAMPAIGN_VERBOSE") == "1":
                print("      fields flagged: %s" % [f for f, idx in bad.items() if i in idx.tolist()])
            if gexp is not None and os.environ.get("CAMPAIGN_VERBOSE") == "1":
                for g in range(gexp.shape[1]):
                    a, b = got.groups[i][g], gexp[i][g]
                    if a["n_alt"] != b["n_alt"] or a["alt"].tolist() != b["alt"].tolist() or not np.allclose(a["af"], b["af"], rtol=1e-6, atol=0, equal_nan=True):
                        print("      group %d depth %d: got n_alt %d alt %s af %s | exp n_alt %d alt %s af %s" % (
                            g, b["total_depth"], a["n_alt"], a["alt"].tolist(), a["af"].tolist(), b["n_alt"], b["alt"].tolist(), b["af"].tolist()))
            print("   tie-excused site %d: depth %s total %d margin %.3g chi2 %.6g got alt %s exp alt %s%s" % (
                i, exp["depth"][i].tolist(), exp["total_depth"][i], margins[i], exp_r["chi2"][i],
                got.sites["alt"][i][:got.sites["n_alt"][i]].tolist(), exp["alt"][i][:exp["n_alt"][i]].tolist(), gd))
        bad = {f: idx[~amb[idx]] for f, idx in bad.items()}
        bad = {f: idx for f, idx in bad.items() if idx.size}
        nvar = int(((exp["status"] & 2) != 0).sum())
        tot_sites += sites; tot_var += nvar; tot_amb += len(excused); tot_bad += sum(len(v) for v in bad.values())
        tot_exc_site += len(exc_site); tot_exc_group += len(exc_group - exc_site)
        print("slab %2d: %5d sites x %6d samples cov %.2f groups %d -> %4d variant, mismatching fields %d, tie-excused sites %d "
              "(site-level call %d, pop-group calls only %d)" % (
                  it, sites, n, cov, ng, nvar, len(bad), len(excused), len(exc_site), len(exc_group - exc_site)), flush=True)
        if bad and gexp is not None:
            for f, idx in bad.items():
                if f.startswith("group."):
                    for i in idx[:3]:
                        print("   group detail site %d margin %g got=%s exp=%s" % (i, margins[i], got.groups[i].tolist(), gexp[i].tolist()))
        if bad:
            print(describe(bad, got.sites, exp))
    print("TOTAL: %d sites (%d variant) against %s in %.0f s: %d mismatches, %d tie-excused sites (site-level call %d, "
          "pop-group calls only %d)" % (
              tot_sites, tot_var, "the real reference" if use_ref else "the restatement", time.time() - t0, tot_bad, tot_amb,
              tot_exc_site, tot_exc_group))
    sys.exit(1 if tot_bad else 0)


if __name__ == "__main__":
    main()
