"""SURVEY section 8 row f2 -- the pileup (BAM -> reference-format batchfile), host-only.
basevar_amd/host/{bamio,pileup}.hpp via the bv_pileup tool against an independent pure-Python
BAM reader + pileup (tests/bam_py.py) on the reference's own BAM fixture and on synthetic BAMs,
indexed (BAI) against linear scans, and the row count the real reference binary produced for the
fixture (2 x range.bam, CHROMOSOME_I:900-1200, --mapq=10 -> 207 CVG rows)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import bam_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
TOOL = os.path.join(ROOT, "basevar_amd", "lib", "bv_pileup")


@pytest.fixture(scope="module")
def tool():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "basevar_amd", "csrc"), "../lib/bv_pileup"], check=True)
    return TOOL


def run_tool(tool, out, fasta, region, bams, mapq=10, extra=()):
    cmd = [tool, "-R", fasta, "--regions", region, "--mapq", str(mapq), "-o", out]
    for b in bams:
        cmd += ["-I", b]
    p = subprocess.run(cmd + list(extra), capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return (gzip.open(out, "rt") if out.endswith(".gz") else open(out)).read()


@pytest.mark.parametrize("region,mapq", [("CHROMOSOME_I:900-1200", 10), ("CHROMOSOME_I:1-2500", 0), ("CHROMOSOME_I:1000-1100", 30),
                                         ("CHROMOSOME_II:1-3000", 10), ("CHROMOSOME_V:1-1500", 10)])
def test_reference_bam_fixture_matches_python_derivation(tool, tmp_path, region, mapq):
    bam = os.path.join(DATA, "range.bam")
    fa = os.path.join(DATA, "ce.fa.gz")
    got = run_tool(tool, str(tmp_path / "a.bf"), fa, region, [bam, bam], mapq)
    lin = run_tool(tool, str(tmp_path / "b.bf.gz"), fa, region, [bam, bam], mapq, extra=["--no-index"])
    assert got == lin  # BAI query == linear scan (and gzip output == plain)
    chrom, span = region.rsplit(":", 1)
    beg, end = map(int, span.split("-"))
    exp = bam_py.batchfile_text([bam, bam], fa, chrom, beg, end, mapq)
    assert got == exp
    # the rows do not depend on how the region is cut into pileup windows, nor on the worker threads
    for w in ("1", "7", "64"):
        assert run_tool(tool, str(tmp_path / "w.bf"), fa, region, [bam, bam], mapq, extra=["--window", w, "--thread", "3"]) == got
    rows = [l.split("\t") for l in got.splitlines()[3:]]
    assert len(rows) == end - beg + 1 and all(len(r) == 9 for r in rows)
    if region == "CHROMOSOME_I:900-1200" and mapq == 10:
        # what the real `basevar basetype` printed for this command: one CVG row per covered position
        assert sum(int(r[3]) > 0 for r in rows) == 207
        assert got.splitlines()[1] == "##SampleIDs=ERS225193,ERS225193"


def _random_reads(rng, n_reads, contig_len, lo, hi):
    reads = []
    for _ in range(n_reads):
        pos = int(rng.integers(lo, hi))
        ops = []
        shape = rng.integers(0, 10)
        if shape == 0: ops = [(4, 5), (1, 3), (0, 20)]                      # 5S3I20M: insertion before any match
        elif shape == 1: ops = [(1, 2), (0, 15), (2, 4), (0, 10)]           # 2I15M4D10M
        elif shape == 2: ops = [(0, 12), (1, 1), (0, 12), (3, 50), (0, 12)]  # with an N skip
        elif shape == 3: ops = [(5, 4), (7, 10), (8, 2), (7, 10), (5, 3)]   # H = X = H
        elif shape == 4: ops = [(0, 10), (6, 2), (0, 10), (4, 6)]           # pad, trailing soft clip
        elif shape == 5: ops = [(0, 30), (2, 1), (0, 30), (1, 5), (0, 30)]
        elif shape == 6: ops = [(4, 3), (2, 2), (0, 25)]                    # deletion before any match (its anchor is free)
        else: ops = [(0, int(rng.integers(20, 101)))]
        qlen = sum(ln for op, ln in ops if op in (0, 1, 4, 7, 8))
        flag = int(rng.choice([0, 16, 0, 16, 1024, 512, 4, 16 | 1, 0, 16]))
        reads.append(dict(tid=0, pos=pos, mapq=int(rng.choice([0, 5, 10, 23, 37, 60, 60, 60])), flag=flag, cigar=ops,
                          seq="".join(rng.choice(list("ACGTN"), qlen, p=[.24, .24, .24, .24, .04])),
                          qual=[int(x) for x in rng.integers(2, 42, qlen)]))
    reads = [r for r in reads if bam_py.end_pos(r) + 10 < contig_len]
    reads.sort(key=lambda r: r["pos"])
    return reads


def _write_fasta(path, name, seq):
    with open(path, "w") as f:
        f.write(">other some description\nACGTACGTAC\n>%s len=%d\n" % (name, len(seq)))
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + "\n")
        f.write(">tail\nGGGG\n")


def test_synthetic_bams_with_indels_clips_and_flags(tool, tmp_path):
    rng = np.random.default_rng(20240607)
    L = 6000
    seq = "".join(rng.choice(list("ACGTacgtN"), L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]))
    fa = str(tmp_path / "syn.fa")
    _write_fasta(fa, "chrS", seq)
    bams = []
    for s in range(3):
        p = str(tmp_path / ("s%d.bam" % s))
        bam_py.write_bam(p, [("chrS", L), ("chrT", 500)], _random_reads(rng, 400, L, 50, 5000),
                         header_text="@HD\tVN:1.6\n@SQ\tSN:chrS\tLN:%d\n@SQ\tSN:chrT\tLN:500\n@RG\tID:a\tPL:x\n@RG\tID:b\tSM:smp%d\tLB:l\n" % (L, s),
                         block_payload=7001)  # records straddle BGZF blocks
        bams.append(p)
    for region, mapq in (("chrS:300-4800", 10), ("chrS:2000-2001", 37), ("chrS:1-5500", 0)):
        got = run_tool(tool, str(tmp_path / "o.bf"), fa, region, bams, mapq)
        beg, end = map(int, region.split(":")[1].split("-"))
        exp = bam_py.batchfile_text(bams, fa, "chrS", beg, end, mapq)
        assert got == exp
    assert got.splitlines()[1] == "##SampleIDs=smp0,smp1,smp2"
    toks = {t for l in exp.splitlines()[3:] for t in l.split("\t")[5].split(" ")}
    assert any(t.startswith("+") for t in toks) and any(t.startswith("-") for t in toks)  # indel tokens do occur


def test_subregion_boundary_of_500kb(tool, tmp_path):
    """__create_a_batchfile walks the region in 500,000-base steps from its start; reads and an insertion that
    straddle the step boundary are piled up once per step, each step with its own first-read-wins map."""
    rng = np.random.default_rng(7)
    L = 500400
    seq = "".join(rng.choice(list("ACGT"), L))
    fa = str(tmp_path / "big.fa")
    _write_fasta(fa, "chrB", seq)
    reads = _random_reads(rng, 60, L, 499850, 500100)
    reads.append(dict(tid=0, pos=499989, mapq=60, flag=0, cigar=[(0, 11), (1, 2), (0, 20)], seq="A" * 33, qual=[30] * 33))  # I at the boundary
    reads.append(dict(tid=0, pos=499979, mapq=60, flag=16, cigar=[(0, 21), (2, 3), (0, 20)], seq="C" * 41, qual=[20] * 41))  # D at the boundary
    reads.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "b.bam")
    bam_py.write_bam(bam, [("chrB", L)], reads)
    got = run_tool(tool, str(tmp_path / "o.bf"), fa, "chrB:1-500300", [bam], 0)
    exp = bam_py.batchfile_text([bam], fa, "chrB", 1, 500300, 0)
    assert got == exp


REF_BAMS = "/root/reference/tests/data/140k_thalassemia_brca_bam"


@pytest.mark.skipif(not os.path.isdir(REF_BAMS), reason="the reference's 100-BAM set is only mounted in the build container")
def test_reference_100_bam_set_reproduces_the_real_binarys_record_count(tool, tmp_path):
    """tests/data/140k_thalassemia_brca_bam/work.log.sh:8 (100 BAMs, chr11:5246595-5248428 + chr17:41197764-41276135,
    --mapq=10) against a surrogate all-'N' FASTA: the real `basevar basetype` wrote 71,984 VCF records (SURVEY.md
    section 8c).  With REF = N every position that carries at least one A/C/G/T call is a record, so the pileup
    alone decides that number."""
    fa = str(tmp_path / "nn.fa.gz")
    with gzip.open(fa, "wt", compresslevel=1) as f:
        for name, L in (("chr11", 135006516), ("chr17", 81195210)):
            f.write(">%s\n" % name)
            full, rem = divmod(L, 60)
            line = "N" * 60 + "\n"
            for _ in range(full // 100000):
                f.write(line * 100000)
            f.write(line * (full % 100000) + ("N" * rem + "\n" if rem else ""))
    bams = [os.path.join(REF_BAMS, l.split()[0]) for l in open(os.path.join(REF_BAMS, "bam100.list")) if l.strip()]
    assert len(bams) == 100
    covered = with_call = 0
    for region in ("chr11:5246595-5248428", "chr17:41197764-41276135"):
        text = run_tool(tool, str(tmp_path / "o.bf"), fa, region, bams, 10)
        rows = [l.split("\t") for l in text.splitlines()[3:]]
        covered += sum(int(r[3]) > 0 for r in rows)
        with_call += sum(any(t in ("A", "C", "G", "T") for t in r[5].split(" ")) for r in rows)
        if region.startswith("chr11"):  # full text against the independent derivation on the short region
            assert text == bam_py.batchfile_text(bams, fa, "chr11", 5246595, 5248428, 10)
    assert (covered, with_call) == (71985, 71984)


def test_cram_and_garbage_inputs_fail_with_a_message(tool, tmp_path):
    fa = os.path.join(DATA, "ce.fa.gz")
    cram = str(tmp_path / "x.cram")
    open(cram, "wb").write(b"CRAM\x03\x00" + b"\0" * 64)
    p = subprocess.run([tool, "-R", fa, "--regions", "CHROMOSOME_I:1-10", "-I", cram, "-o", str(tmp_path / "o.bf")], capture_output=True, text=True)
    assert p.returncode != 0 and "CRAM" in p.stderr
    junk = str(tmp_path / "junk.bam")
    open(junk, "wb").write(b"hello world, not a bam file at all")
    p = subprocess.run([tool, "-R", fa, "--regions", "CHROMOSOME_I:1-10", "-I", junk, "-o", str(tmp_path / "o.bf")], capture_output=True, text=True)
    assert p.returncode != 0 and "BGZF" in p.stderr
    p = subprocess.run([tool, "-R", fa, "--regions", "NOPE:1-10", "-I", os.path.join(DATA, "range.bam"), "-o", str(tmp_path / "o.bf")],
                       capture_output=True, text=True)
    assert p.returncode != 0 and "not found" in p.stderr
