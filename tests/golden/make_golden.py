#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REAL reference.

Run in the build container (where /root/reference exists):

    make -C oracle ref && python tests/golden/make_golden.py

Each fixture is data only: the seeded input planes and the per-site records that the
reference's own code (compiled unmodified into oracle/_ref/libbvref.so, see oracle/Makefile
and oracle/ref_driver.cpp) produced for them.  The reference has no golden outputs of its
own for this path (tests/io/test_algorithm.cpp:41: "how to test EM?"); its test *inputs*
(test_algorithm.cpp:13-31) are evaluated here too and stored in known_answers.json.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from basevar_amd.synth import make_slab  # noqa: E402

A, Cc, G, T, N, INS, DEL, REV = 0, 1, 2, 3, 8, 9, 10, 4


def edge_slab():
    """Hand-built sites: the reference's observable corner cases."""
    n = 16
    sites = []

    def site(ref, cells):
        bs = np.full(n, N, np.uint8); q = np.zeros(n, np.uint8); mq = np.zeros(n, np.uint8); rp = np.zeros(n, np.uint16)
        for i, c in enumerate(cells):
            b, ph = c[0], c[1]
            rev = c[2] if len(c) > 2 else 0
            bs[i] = b | (REV if (rev and b < 8) else 0)
            q[i] = ph
            mq[i] = c[3] if len(c) > 3 else (60 if b != N else 0)
            rp[i] = c[4] if len(c) > 4 else (i + 1 if b != N else 0)
        sites.append((ref, bs, q, mq, rp))

    site(G, [(Cc, 30), (T, 30)])                       # exact tie: only C reported, AF 1
    site(G, [(T, 30), (Cc, 30)])                       # same with the order swapped
    site(G, [(A, 30), (Cc, 30), (T, 30)])              # three-way tie
    site(A, [(A, 30), (A, 30), (G, 30), (G, 35)])      # ref given as 'a' upstream -> toupper
    site(A, [(G, 30)] * 11)                            # mono-allelic, depth 11 -> QUAL 5000
    site(A, [(G, 30)] * 10)                            # depth 10 -> QUAL 0
    site(4, [(A, 30), (A, 30)])                        # ref 'N'
    site(Cc, [(INS, 20), (N, 0), (DEL, 25)])           # only indel / N tokens: depth 0, no call
    site(T, [])                                        # nothing covered at all
    # strand-bias / rank-sum example of SURVEY 8a: ref A; bases A A G N G A + C A G
    bases = [A, A, G, N, G, A, INS, Cc, A, G]
    strands = [0, 1, 0, 0, 1, 1, 0, 0, 0, 0]
    mapq = [60, 50, 60, 0, 30, 60, 60, 20, 60, 40]
    rank = [5, 10, 5, 0, 30, 12, 7, 7, 20, 1]
    bq = [ord(c) - 33 for c in "I5I!?II+I5"]
    site(A, [(b, bq[i], strands[i], mapq[i], rank[i]) for i, b in enumerate(bases)])
    site(A, [(A, 40)] * 8 + [(G, 40, 1)] * 8)          # every cell covered, perfectly strand-biased ALT
    site(Cc, [(A, 2), (Cc, 2), (G, 2), (T, 2)] * 4)    # phred 2 everywhere: all four bases active
    site(G, [(G, 93)] * 3 + [(T, 93)] * 2)             # maximum phred
    site(T, [(T, 35, 0, 60, 300), (T, 35, 1, 60, 2000), (A, 35, 0, 20, 1500), (A, 30, 1, 10, 65535)])  # long reads
    site(A, [(A, 0)])                                  # phred 0, reference base only
    site(Cc, [(A, 0)])                                 # phred 0 ALT: the reference reports AF = NaN
    site(A, [(A, 30)] * 15 + [(Cc, 10)])               # weak alt below the LRT threshold
    site(4, [(A, 35), (Cc, 35, 1), (G, 35), (T, 35, 1)] * 4)  # ref 'N', four supported bases: FOUR alts
    S = len(sites)
    slab = {
        "n_sites": S, "n_samples": n, "pitch": n, "n_groups": 2,
        "base_strand": np.stack([s[1] for s in sites]), "qual": np.stack([s[2] for s in sites]),
        "mapq": np.stack([s[3] for s in sites]), "rpr": np.stack([s[4] for s in sites]),
        "ref_base": np.array([s[0] for s in sites], np.uint8),
        "group_id": np.array([0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 1, 0xFF, 0xFF, 0, 1], np.uint8),
    }
    return slab


def deep_sor_slab():
    """Very deep sites whose strand-count products pass 2^31: pins what the compiled reference does
    with the `int` overflow of basetype.cpp:286.  Run-structured so that the fixture stays small."""
    n = 260000
    sites = []
    #            A+      A-      G+      G-     (ref A, alt G; rest uncovered)
    layouts = [(60000, 61000, 50000, 52000), (90000, 47000, 46400, 70000), (120000, 20000, 18000, 100000),
               (46341, 46341, 46341, 46341), (130000, 129000, 400, 300)]
    for l in layouts:
        bs = np.full(n, N, np.uint8); q = np.zeros(n, np.uint8); mq = np.zeros(n, np.uint8); rp = np.zeros(n, np.uint16)
        o = 0
        for k, cnt in enumerate(l):
            code = (A if k < 2 else G) | (REV if k & 1 else 0)
            bs[o:o + cnt] = code
            q[o:o + cnt] = 30 + (k % 3)
            mq[o:o + cnt] = 60 - k
            rp[o:o + cnt] = 10 + 5 * k
            o += cnt
        sites.append((A, bs, q, mq, rp))
    return {"n_sites": len(sites), "n_samples": n, "pitch": n, "n_groups": 0,
            "base_strand": np.stack([s[1] for s in sites]), "qual": np.stack([s[2] for s in sites]),
            "mapq": np.stack([s[3] for s in sites]), "rpr": np.stack([s[4] for s in sites]),
            "ref_base": np.array([s[0] for s in sites], np.uint8)}


def real_bam_slab():
    """Real sequencing data: the reference's own 100-BAM test set (tests/data/140k_thalassemia_brca_bam, work.log.sh:8),
    piled up by basevar_amd/lib/bv_pileup over chr11:5246595-5248428 and the first 6 kb of the chr17 region, covered
    positions only.  hg19 is not shipped with the reference, so REF is the majority call of the position (ties: lowest
    base code) -- most sites hom-ref, the rest real SNVs/errors -- and every 37th site gets REF 'N'.  Pop-groups come from
    the set's sample_group.info."""
    import gzip
    import subprocess
    import tempfile
    bdir = "/root/reference/tests/data/140k_thalassemia_brca_bam"
    tool = os.path.join(ROOT, "basevar_amd", "lib", "bv_pileup")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "basevar_amd", "csrc"), "../lib/bv_pileup"], check=True)
    bams = [os.path.join(bdir, l.split()[0]) for l in open(os.path.join(bdir, "bam100.list")) if l.strip()]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "nn.fa.gz")
        with gzip.open(fa, "wt", compresslevel=1) as f:
            for name, L in (("chr11", 5250000), ("chr17", 41210000)):
                f.write(">%s\n" % name + ("N" * 60 + "\n") * (L // 60 + 1))
        ids = None
        for region in ("chr11:5246595-5248428", "chr17:41197764-41203763"):
            out = os.path.join(tmp, "o.bf")
            cmd = [tool, "-R", fa, "--regions", region, "--mapq", "10", "-o", out]
            for b in bams:
                cmd += ["-I", b]
            subprocess.run(cmd, check=True, capture_output=True)
            lines = open(out).read().splitlines()
            ids = lines[1].split("=", 1)[1].split(",")
            rows += [l.split("\t") for l in lines[3:] if int(l.split("\t")[3]) > 0]
    n = len(ids)
    S = len(rows)
    pitch = (n + 15) // 16 * 16  # row pitch of the planes: a multiple of 16 bytes
    bs = np.full((S, pitch), N, np.uint8); q = np.zeros((S, pitch), np.uint8); mq = np.zeros((S, pitch), np.uint8)
    rp = np.zeros((S, pitch), np.uint16); ref = np.zeros(S, np.uint8)
    for s, r in enumerate(rows):
        toks, quals, strands = r[5].split(" "), r[6].split(" "), r[8].split(" ")
        mqs, ranks = r[4].split(" "), r[7].split(" ")
        cnt = [0, 0, 0, 0]
        for i, t in enumerate(toks):
            if t[0] == "N":
                continue
            if t[0] in code:
                bs[s, i] = code[t[0]] | (REV if strands[i] == "-" else 0)
                cnt[code[t[0]]] += 1
            else:
                bs[s, i] = INS if t[0] == "+" else DEL
            q[s, i] = ord(quals[i]) - 33
            mq[s, i] = int(mqs[i])
            rp[s, i] = int(ranks[i])
        ref[s] = 4 if s % 37 == 36 else int(np.argmax(cnt))
    s2g = dict(l.split()[:2] for l in open(os.path.join(bdir, "sample_group.info")) if l.strip())
    gnames = sorted(set(s2g.values()))
    gid = np.array([gnames.index(s2g[i]) if i in s2g else 0xFF for i in ids], np.uint8)
    return {"n_sites": S, "n_samples": n, "pitch": pitch, "n_groups": len(gnames), "base_strand": bs, "qual": q, "mapq": mq,
            "rpr": rp, "ref_base": ref, "group_id": gid}


FIXTURES = {
    # name: (slab factory, user min_af)
    "edge16": (edge_slab, 0.01),
    "deep_sor": (deep_sor_slab, 0.01),
    "dense_64x500": (lambda: make_slab(64, 500, seed=11, coverage=0.6, n_groups=2, ref_n_frac=0.05), 0.01),
    "nipt_96x4000": (lambda: make_slab(96, 4000, seed=12, coverage=0.08, n_groups=2), 0.01),
    "ragged_40x1003": (lambda: make_slab(40, 1003, seed=13, coverage=0.25, n_groups=3, pitch=1008), 0.01),
    "real_100bam": (real_bam_slab, 0.05),   # --min-af=0.05 as in the set's work.log.sh
    # the headline row lengths, records from the real reference (the other fixtures stop at 4,000 samples)
    "nipt_20x100000": (lambda: make_slab(20, 100000, seed=15, coverage=0.08, n_groups=2), 0.01),
    "deep_6x1000000": (lambda: make_slab(6, 1000000, seed=16, coverage=0.05, site_offset=14), 0.01),
    "lowq_48x800": (lambda: make_slab(48, 800, seed=14, coverage=0.4, qual_mean=12.0, qual_sd=8.0, qual_min=1,
                                      qual_max=60), 0.01),
}


def main():
    ref = oracle.Reference()
    res = oracle.Restatement()
    for name, (factory, user_af) in FIXTURES.items():
        slab = factory()
        maf = res.min_af(slab["n_samples"], user_af)
        sites, groups = ref.run(slab, maf)
        out = {k: v for k, v in slab.items() if isinstance(v, np.ndarray)}
        out["n_samples"] = np.int64(slab["n_samples"])
        out["n_groups"] = np.int64(slab.get("n_groups", 0))
        out["min_af"] = np.float64(maf)
        out["expected_sites"] = sites.view(np.uint8)
        if groups is not None:
            out["expected_groups"] = groups.view(np.uint8).reshape(groups.shape[0], -1)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-16s sites=%d variants=%d -> %s (%d bytes)" % (
            name, len(sites), int(((sites["status"] & 2) != 0).sum()), os.path.basename(path), os.path.getsize(path)))

    ka = {
        "source": "reference functions of src/algorithm.h called through oracle/_ref (inputs: tests/io/test_algorithm.cpp:13-31 and SURVEY.md section 4)",
        "chi2_test": [[x, 1.0, ref.chi2_test(x, 1.0)] for x in (24.0, 0.0, 3.84, 1500.0, -0.1, 1.0, 30.5, 100.0)],
        "norm_dist": [[x, ref.norm_dist(x)] for x in (1.96, 0.0, 0.5, 5.0, 40.0)],
        "fisher_exact_test": [[list(t), ref.fisher(*t)] for t in
                              [(345, 455, 260, 345), (8, 4, 4, 9), (10, 5, 4, 9), (3, 4, 4, 5), (1, 1, 1, 1),
                               (4000, 4100, 3, 2), (0, 10, 10, 0), (1200, 1300, 900, 700), (50000, 50000, 40000, 41000)]],
        "wilcoxon_ranksum_test": [[[1, 5, 3, 10, 3, 3, 4, 5], [6, 7, 2, 2, 8, 9, 10],
                                   ref.wilcoxon([1, 5, 3, 10, 3, 3, 4, 5], [6, 7, 2, 2, 8, 9, 10])]],
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, default=lambda v: None if v != v else v)
    print("known_answers.json written")


if __name__ == "__main__":
    main()
