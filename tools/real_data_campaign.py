#!/usr/bin/env python3
"""Differential run on REAL sequencing data: every covered position of the reference's own 100-BAM test set
(tests/data/140k_thalassemia_brca_bam/work.log.sh:8: chr11:5246595-5248428 + chr17:41197764-41276135, --mapq=10),
piled up by basevar_amd/lib/bv_pileup, through the engine (C ABI) against the real reference code (oracle/_ref).

    python tools/real_data_campaign.py build     # in the build container (needs /root/reference): writes the slab
                                                 # to tests/golden/_local/real_all.npz (git-ignored, travels with gpurun)
    python tools/real_data_campaign.py           # on the GPU box

REF is not available (hg19 is not shipped): three REF choices are run -- the majority call, 'N' (every position with a
call becomes a record: the real binary wrote 71,984 of them), and the majority call shifted by one
base (every site a multi-allelic variant) -- each at --min-af 0.05 and 0.01, with the set's pop-groups."""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SLAB = os.path.join(ROOT, "tests", "golden", "_local", "real_all.npz")


def build():
    bdir = "/root/reference/tests/data/140k_thalassemia_brca_bam"
    tool = os.path.join(ROOT, "basevar_amd", "lib", "bv_pileup")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "basevar_amd", "csrc"), "../lib/bv_pileup"], check=True)
    bams = [os.path.join(bdir, l.split()[0]) for l in open(os.path.join(bdir, "bam100.list")) if l.strip()]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "nn.fa.gz")
        with gzip.open(fa, "wt", compresslevel=1) as f:
            for name, L in (("chr11", 5250000), ("chr17", 41280000)):
                f.write(">%s\n" % name + ("N" * 60 + "\n") * (L // 60 + 1))
        for region in ("chr11:5246595-5248428", "chr17:41197764-41276135"):
            out = os.path.join(tmp, "o.bf")
            cmd = [tool, "-R", fa, "--regions", region, "--mapq", "10", "-o", out]
            for b in bams:
                cmd += ["-I", b]
            subprocess.run(cmd, check=True, capture_output=True)
            lines = open(out).read().splitlines()
            ids = lines[1].split("=", 1)[1].split(",")
            rows += [l.split("\t") for l in lines[3:] if int(l.split("\t")[3]) > 0]
    n, S = len(ids), len(rows)
    pitch = (n + 15) // 16 * 16
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    bs = np.full((S, pitch), 8, np.uint8); q = np.zeros((S, pitch), np.uint8); mq = np.zeros((S, pitch), np.uint8)
    rp = np.zeros((S, pitch), np.uint16); ref = np.zeros(S, np.uint8)
    for s, r in enumerate(rows):
        toks, quals, strands, mqs, ranks = r[5].split(" "), r[6].split(" "), r[8].split(" "), r[4].split(" "), r[7].split(" ")
        cnt = [0, 0, 0, 0]
        for i, t in enumerate(toks):
            if t[0] == "N":
                continue
            if t[0] in code:
                bs[s, i] = code[t[0]] | (4 if strands[i] == "-" else 0)
                cnt[code[t[0]]] += 1
            else:
                bs[s, i] = 9 if t[0] == "+" else 10
            q[s, i] = ord(quals[i]) - 33; mq[s, i] = int(mqs[i]); rp[s, i] = int(ranks[i])
        ref[s] = int(np.argmax(cnt))
    s2g = dict(l.split()[:2] for l in open(os.path.join(bdir, "sample_group.info")) if l.strip())
    gn = sorted(set(s2g.values()))
    gid = np.array([gn.index(s2g[i]) if i in s2g else 255 for i in ids], np.uint8)
    os.makedirs(os.path.dirname(SLAB), exist_ok=True)
    np.savez_compressed(SLAB, base_strand=bs, qual=q, mapq=mq, rpr=rp, ref_base=ref, group_id=gid, n_samples=n, n_groups=len(gn))
    print("%s: %d covered positions x %d samples, %d pop-groups" % (SLAB, S, n, len(gn)))


def run():
    import basevar_amd
    import oracle
    from parity import ambiguous_sites, compare_groups, compare_sites, describe
    d = np.load(SLAB)
    ref = oracle.Reference(); res = oracle.Restatement()
    tot = bad_tot = amb_tot = 0
    for mode in ("consensus", "refN", "ref_shift", "refN_nogroups"):
        for user_af in (0.05, 0.01):
            slab = {k: d[k] for k in ("base_strand", "qual", "mapq", "rpr", "ref_base", "group_id")}
            slab["n_samples"] = int(d["n_samples"]); slab["n_groups"] = int(d["n_groups"])
            if mode == "refN_nogroups":  # without pop-groups short rows take the persistent pass-2 kernel
                slab.pop("group_id"); slab["n_groups"] = 0
                slab["ref_base"] = np.full_like(d["ref_base"], 4)
            if mode == "refN":
                slab["ref_base"] = np.full_like(d["ref_base"], 4)
            if mode == "ref_shift":
                slab["ref_base"] = (d["ref_base"] + 1) % 4
            maf = res.min_af(slab["n_samples"], user_af)
            eng = basevar_amd.BaseTypeEngine(len(slab["ref_base"]), maf); got = eng.lrt(slab); eng.close()
            exp, gexp = ref.run(slab, maf, n_threads=32)
            exp_r, _, margins = res.run_with_margins(slab, maf, n_threads=32)
            amb = ambiguous_sites(exp_r, margins)
            bad = compare_sites(got.sites, exp, check_chi2=False)
            bad.update(compare_groups(got.groups, gexp, (exp["status"] & 2) != 0))
            exc = set()
            for f, idx in bad.items():
                exc.update(idx[amb[idx]].tolist())
            bad = {f: idx[~amb[idx]] for f, idx in bad.items()}
            bad = {f: i for f, i in bad.items() if i.size}
            nv = int(((exp["status"] & 2) != 0).sum())
            print("real data %-13s min_af %.2f: %d sites (%d variant), mismatching fields %d, tie-excused %d" % (
                mode, user_af, len(exp), nv, sum(len(v) for v in bad.values()), len(exc)))
            if bad:
                print(describe(bad, got.sites, exp)[:2000])
            tot += len(exp); bad_tot += sum(len(v) for v in bad.values()); amb_tot += len(exc)
    print("TOTAL real-data: %d sites against the real reference: %d mismatches, %d tie-excused" % (tot, bad_tot, amb_tot))
    return 1 if bad_tot else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        sys.exit(run())
