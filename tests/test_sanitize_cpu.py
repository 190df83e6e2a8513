"""CPU guards for the threaded host code that parses binary input (sanitizers on the CPU build): the
`sanitize` target of basevar_amd/csrc/Makefile builds AddressSanitizer + UBSan and ThreadSanitizer variants of the batchfile
producer's check, of bv_call's reader + emitter without an engine (tests/cpp/host_fuzz.cpp), of the BGZF / tabix writer and of
bv_pileup.  They must run clean on valid inputs (1 .. 8 threads) and, on a seeded corpus of truncated and byte-flipped BAM, BAI,
FASTA, BGZF, gzip and plain batchfile inputs, end with exit code 0 or with ONE message and exit code 1 -- never a signal, never a
sanitizer report.  (Reference code replaced: src/basetype_caller.cpp:586-611 reader, :800-1101 pileup, :1103-1260 emitter; the
reference itself reads through htslib.)"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from test_host_formats import _write_batchfiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "basevar_amd", "lib", "san")
DATA = os.path.join(ROOT, "tests", "golden", "data")
ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:detect_leaks=1:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98",
           TSAN_OPTIONS="exitcode=97:halt_on_error=1")


@pytest.fixture(scope="module")
def san():
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "basevar_amd", "csrc"), "sanitize"])
    return SAN


def run(cmd, timeout=300):
    p = subprocess.run(cmd, capture_output=True, text=True, env=ENV, timeout=timeout, errors="replace")
    report = [l for l in p.stderr.splitlines() if "Sanitizer" in l or "runtime error:" in l]
    assert not report, "%s\n%s" % (" ".join(cmd), p.stderr[-3000:])
    assert p.returncode in (0, 1), "exit code %d of %s\n%s" % (p.returncode, " ".join(cmd), p.stderr[-2000:])  # (negative: a signal)
    if p.returncode == 1:
        assert p.stderr.strip(), "a failure without a message: %s" % " ".join(cmd)
    return p


def damaged(path, out_dir, seed, n_each=6):
    """seeded variants of a file: truncated at random lengths (0 and header-only lengths among them), single and burst byte
    flips, a zeroed block, a doubled tail"""
    raw = open(path, "rb").read()
    rng = np.random.default_rng(seed)
    base = os.path.basename(path)
    out = []

    def put(tag, data):
        q = os.path.join(out_dir, "%s.%s" % (tag, base))
        open(q, "wb").write(bytes(data))
        out.append(q)
    for k, cut in enumerate([0, 1, 17, 28] + [int(x) for x in rng.integers(29, max(30, len(raw)), size=n_each)]):
        put("trunc%d" % k, raw[:min(cut, len(raw))])
    for k in range(n_each):
        b = bytearray(raw)
        for _ in range(1 if k % 2 == 0 else 24):
            i = int(rng.integers(0, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        put("flip%d" % k, b)
    b = bytearray(raw)
    i = int(rng.integers(0, max(1, len(b) - 600)))
    b[i:i + 512] = bytes(512)
    put("zeroed", b)
    put("doubled", raw + raw[len(raw) // 2:])
    return out


def test_sanitized_producer_and_emitter_on_valid_inputs(san, tmp_path):
    d = tmp_path / "ok"; d.mkdir()
    files = _write_batchfiles(str(d), 5, 37, 500, seed=11, plain_last=True)   # BGZF, gzip and plain text among them
    for kind in ("asan", "tsan"):
        p = run([os.path.join(san, "producer_check." + kind), ",".join(files)])  # 1 / 2 / 3 / 8 threads inside
        assert p.returncode == 0 and p.stdout.startswith("OK "), p.stdout + p.stderr
        for t in (1, 3, 8):
            p = run([os.path.join(san, "host_fuzz." + kind), str(t), ",".join(files)])
            assert p.returncode == 0 and p.stdout.startswith("OK "), p.stdout + p.stderr
    p = run([os.path.join(san, "bgzf_tabix_check.asan"), str(tmp_path / "t.tsv.gz"), "3"])
    assert p.returncode == 0


def test_sanitized_reader_on_damaged_batchfiles(san, tmp_path):
    """truncated / byte-flipped BGZF, gzip and plain-text batchfiles beside intact ones: a message or a clean end, never a signal"""
    d = tmp_path / "src"; d.mkdir()
    files = _write_batchfiles(str(d), 3, 23, 260, seed=12, plain_last=True)  # bf_00.gz BGZF, bf_01.gz gzip, bf_02.txt plain
    n_msg = n_ok = 0
    for victim in range(3):
        c = tmp_path / ("corpus%d" % victim); c.mkdir()
        for q in damaged(files[victim], str(c), seed=100 + victim):
            mix = list(files)
            mix[victim] = q
            for kind, t in (("asan", 3), ("tsan", 4)):
                p = run([os.path.join(san, "host_fuzz." + kind), str(t), ",".join(mix)])
                n_msg += p.returncode == 1
                n_ok += p.returncode == 0
    assert n_msg > 20 and n_ok > 5   # (a flipped phred byte is still a valid file; a truncated member is not)


def test_sanitized_pileup_on_damaged_bam_bai_fasta(san, tmp_path):
    """bv_pileup on the reference's own BAM fixture (tests/data/range.bam): intact, then with a damaged BAM, BAI or FASTA"""
    work = tmp_path / "w"; work.mkdir()
    for f in ("range.bam", "range.bam.bai", "ce.fa.gz"):
        shutil.copy(os.path.join(DATA, f), str(work / f))
    base = ["-R", str(work / "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200", "--mapq", "10", "-o", str(tmp_path / "out.bf.gz")]
    for kind in ("asan", "tsan"):
        p = run([os.path.join(san, "bv_pileup." + kind)] + base + ["-I", str(work / "range.bam"), "-I", str(work / "range.bam"), "--thread", "3"])
        assert p.returncode == 0, p.stderr[-1500:]
    n_msg = 0
    for name, seed in (("range.bam", 1), ("range.bam.bai", 2), ("ce.fa.gz", 3)):
        c = tmp_path / ("c_" + name); c.mkdir()
        for q in damaged(os.path.join(DATA, name), str(c), seed=seed, n_each=5):
            w2 = tmp_path / "w2"
            if w2.exists():
                shutil.rmtree(str(w2))
            w2.mkdir()
            for f in ("range.bam", "range.bam.bai", "ce.fa.gz"):
                shutil.copy(q if f == name else os.path.join(DATA, f), str(w2 / f))
            args = ["-R", str(w2 / "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200", "--mapq", "10", "-o", str(tmp_path / "o2.bf.gz"), "-I", str(w2 / "range.bam")]
            p = run([os.path.join(san, "bv_pileup.asan")] + args + ["--thread", "2"])
            n_msg += p.returncode == 1
            if name == "range.bam.bai":  # without the index the BAM is scanned: must not depend on the damaged file at all
                p = run([os.path.join(san, "bv_pileup.asan")] + args + ["--no-index"])
                assert p.returncode == 0
    assert n_msg > 10
