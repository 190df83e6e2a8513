#!/usr/bin/env python3
"""Randomised text-level campaign (GPU box): reference-format batchfiles of random shape -> `bv_call` (reader, GPU engine,
pop-groups, VCF / CVG emit) against the REFERENCE's own per-position caller on the same rows (`_basevar_caller`, compiled where
it lies: oracle/_ref/libbvcaller.so, tests/ref_caller.py).  CVG lines must be byte-identical; VCF lines byte-identical, or equal
field by field to 1e-6 where a deep site's float rounds differently in its last printed digit.

    python3 tools/text_campaign.py [rounds]          TEXT_CAMPAIGN_SEED=<n>  TEXT_CAMPAIGN_DEEP=1 (rows of 4,000-15,000 samples) | 2 (50,000-72,000)
"""
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_host_formats as T  # noqa: E402  (the generator of reference-format batchfiles and the reference-caller helper)
import oracle  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(os.environ.get("TEXT_CAMPAIGN_SEED", "1"))
    rng = np.random.default_rng(seed)
    restatement = oracle.Restatement()
    tmp = tempfile.mkdtemp(prefix="bv_text_campaign_")
    exe = T.cxx(os.path.join(ROOT, "basevar_amd", "host", "bv_call.cpp"), os.path.join(tmp, "bv_call"), ["-lz"])
    t0 = time.time()
    tot_cvg = tot_vcf = same_vcf = 0
    for r in range(rounds):
        n_files = int(rng.integers(1, 7))
        # TEXT_CAMPAIGN_DEEP=1: rows of 4,000-15,000 samples (the fused short-row kernel's lengths), fewer positions per round
        deep = os.environ.get("TEXT_CAMPAIGN_DEEP") == "1"
        per = int(rng.choice([1000, 1700, 2500])) if deep else int(rng.choice([3, 8, 20, 64, 150, 400]))
        if os.environ.get("TEXT_CAMPAIGN_DEEP") == "2":  # rows of 50,000-72,000 samples: the long-row kernels
            per, deep = int(rng.choice([10000, 12000])), True
        if deep:
            n_files = int(rng.integers(5 if per >= 10000 else 4, 7))
        n_samples = n_files * per
        n_sites = int(rng.integers(20, 70)) if not deep else (int(rng.integers(6, 11)) if per >= 10000 else int(rng.integers(12, 25)))
        n_groups = int(rng.choice([0, 0, 1, 2, 5, 9]))
        d = os.path.join(tmp, "r%d" % r)
        os.makedirs(d)

        class P:  # (make_batchfiles wants a pathlib-like tmp_path)
            def __truediv__(self, name):
                return os.path.join(d, name)
        paths, ids, sites = T.make_batchfiles(P(), n_sites=n_sites, n_samples=n_samples, n_files=n_files, seed=int(rng.integers(1 << 30)))
        grp = {}
        args = [exe, "--batchfiles", ",".join(paths), "--output-vcf", os.path.join(d, "o.vcf"), "--output-cvg", os.path.join(d, "o.cvg"),
                "--batch-sites", str(int(rng.choice([7, 32, 4096]))), "--thread", str(int(rng.integers(1, 6)))]
        if n_groups:
            names = ["g%02d" % g for g in range(n_groups)]
            assign = rng.integers(-1, n_groups, size=n_samples)
            grp = {names[g]: [int(i) for i in np.nonzero(assign == g)[0]] for g in range(n_groups) if (assign == g).any()}
            pf = os.path.join(d, "groups.info")
            with open(pf, "w") as fh:
                for g, idx in grp.items():
                    for i in idx:
                        fh.write("%s\t%s\n" % (ids[i], g))
            args += ["--pop-group", pf]
        subprocess.check_call(args, stderr=subprocess.DEVNULL)
        ref = T.reference_caller_lines(paths, n_samples, restatement.min_af(n_samples, 0.01), grp)
        assert ref is not None, "oracle/_ref/libbvcaller.so is needed"
        got_cvg = [l for l in open(os.path.join(d, "o.cvg")).read().split("\n") if l and not l.startswith("#")]
        got_vcf = [l for l in open(os.path.join(d, "o.vcf")).read().split("\n") if l and not l.startswith("#")]
        assert got_cvg == ref[0], ("CVG", r, n_samples, [(a, b) for a, b in zip(got_cvg, ref[0]) if a != b][:2])
        assert len(got_vcf) == len(ref[1]), ("VCF count", r, len(got_vcf), len(ref[1]))
        for a, b in zip(got_vcf, ref[1]):
            if a == b:
                same_vcf += 1
                continue
            fa, fb = re.split("[\t;,=:]", a), re.split("[\t;,=:]", b)
            assert len(fa) == len(fb), (r, a[:300], b[:300])
            for x, y in zip(fa, fb):
                if x != y:
                    assert abs(float(x) - float(y)) <= 1e-6 * max(1.0, abs(float(y))) + 1.5e-6, (r, n_samples, x, y, a[:200])
        tot_cvg += len(got_cvg)
        tot_vcf += len(got_vcf)
        print("round %d: %d files x %d samples, %d sites, %d groups: %d CVG lines identical, %d VCF lines" %
              (r, n_files, per, n_sites, len(grp), len(got_cvg), len(got_vcf)), flush=True)
    print("TOTAL: %d rounds against the reference's own per-position caller in %.0f s: %d CVG lines byte-identical, %d VCF lines "
          "(%d byte-identical, %d equal to 1e-6 per field), 0 mismatches" % (rounds, time.time() - t0, tot_cvg, tot_vcf, same_vcf, tot_vcf - same_vcf))


if __name__ == "__main__":
    main()
