"""Rows f1 / f3 of SURVEY.md section 8: the batchfile reader and the VCF/CVG emitter
(basevar_amd/host/batchfile.hpp, vcf_emit.hpp, bv_call.cpp).

CPU part: the C++ harness tests/cpp/host_formats_check.cpp checks the tokenisers/formatters
against the reference's own ngslib::split/join (through oracle/_ref), round-trips batchfile rows,
and prints CVG/VCF lines for records computed by the oracle restatement; this file re-derives
every line independently in Python from the same records (double entry).
GPU part: bv_call end to end on gzip batchfiles against lines derived from the oracle's records.
"""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "basevar_amd", "lib")
BASES = "ACGT"


def cxx(src, exe, extra=()):
    import __graft_entry__ as g
    g.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-L", LIB,
                           "-lbasevar_amd", "-Wl,-rpath," + LIB, "-pthread", "-o", exe] + list(extra))
    return exe


def f6(x):
    """std::to_string(double): printf("%f")."""
    return "%f" % x


def g6(x):
    """ostringstream << double with default precision: printf("%g")."""
    return "%g" % x


def expected_cvg(site, r):
    if r["total_depth"] == 0:
        return None
    indel = {}
    for t in site["bases"]:
        if t[0] in "N" + BASES:
            continue
        indel[t] = indel.get(t, 0) + 1
    ind = ",".join("%s|%d" % (k, indel[k]) for k in sorted(indel)) if indel else "."
    sb = r["cvg_sb"]
    return "\t".join([site["chrom"], str(site["pos"]), site["ref"], str(int(r["total_depth"]))] +
                     [str(int(d)) for d in r["depth"]] +
                     [ind, f6(r["cvg_fs"]), f6(r["cvg_sor"]), "%d,%d,%d,%d" % tuple(int(x) for x in sb)])


def expected_vcf(site, r, groups, gnames):
    n_alt = int(r["n_alt"])
    if n_alt == 0:
        return None
    alts = [BASES[b] for b in r["alt"][:n_alt]]
    gt = {b: "./%d" % (i + 1) for i, b in enumerate(alts)}
    up = site["ref"][0].upper()
    samples = []
    for tok, qc, st in zip(site["bases"], site["quals"], site["strands"]):
        fb = tok[0]
        if fb in "N+-":
            samples.append("./.")
            continue
        if fb not in gt:
            gt[fb] = "./."
        g = "0/." if fb == up else gt[fb]
        bp = 1.0 - np.exp((ord(qc) - 33) * -0.23025850929940458)
        samples.append("%s:%s:%s:%s" % (g, fb, st, f6(bp)))
    info = ["CM_DP=%d" % r["total_depth"], "CM_AC=" + ",".join(str(int(r["depth"][b])) for b in r["alt"][:n_alt]),
            "CM_AF=" + ",".join(g6(x) for x in r["af"][:n_alt]), "CM_CAF=" + ",".join(g6(x) for x in r["caf"][:n_alt]),
            "MQRankSum=%d" % int(r["mq_ranksum"]), "ReadPosRankSum=%d" % int(r["rpr_ranksum"]),
            "BaseQRankSum=%d" % int(r["bq_ranksum"]), "QD=" + f6(r["qd"]), "SOR=" + f6(r["var_sor"]), "FS=" + f6(r["var_fs"]),
            "SB_REF=%d,%d" % (r["var_sb"][0], r["var_sb"][1]), "SB_ALT=%d,%d" % (r["var_sb"][2], r["var_sb"][3])]
    if groups is not None:
        for gname, gr in zip(gnames, groups):
            if gr["n_alt"]:
                info.append(gname + "_AF=" + ",".join(g6(x) for x in gr["af"][:gr["n_alt"]]))
    flt = "." if r["qual"] > 20 else "LowQual"
    return "\t".join([site["chrom"], str(site["pos"]), ".", site["ref"], ",".join(alts), f6(r["qual"]), flt, ";".join(info),
                      "GT:AB:SO:BP"] + samples)


def _write_batchfiles(dirpath, n_files, per, sites, seed, plain_last=False, bgzf=True):
    """small reference-format batchfiles (gzip; optionally the last one plain text): `sites` = list of row counts per file, or int"""
    import gzip
    rng = np.random.default_rng(seed)
    paths = []
    rows_of = sites if isinstance(sites, (list, tuple)) else [sites] * n_files
    refs = rng.integers(0, 4, size=max(rows_of))
    for f in range(n_files):
        lines = ["##fileformat=BaseVarBatchFile_v1.0", "##SampleIDs=" + ",".join("S%d" % (f * per + i) for i in range(per)),
                 "#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\tReadbases\tReadbasesQuality\tReadPositionRank\tStrand"]
        for s in range(rows_of[f]):
            cov = rng.random(per) < (0.0 if s % 17 == 5 else 0.3)   # every 17th position is covered by nobody (dropped rows)
            toks = [[], [], [], [], []]
            for i in range(per):
                if cov[i]:
                    b = "ACGT"[int(rng.integers(0, 4))]
                    if rng.random() < 0.05:
                        b = ("+" if rng.random() < 0.5 else "-") + b + "TT"
                    toks[0].append(str(int(rng.integers(0, 61)))); toks[1].append(b); toks[2].append(chr(33 + int(rng.integers(2, 42))))
                    toks[3].append(str(int(rng.integers(1, 151)))); toks[4].append("+-"[int(rng.integers(0, 2))])
                else:
                    toks[0].append("0"); toks[1].append("N"); toks[2].append("!"); toks[3].append("0"); toks[4].append(".")
            lines.append("\t".join(["chr1", str(100 + s), "ACGT"[refs[s]], str(int(cov.sum()))] + [" ".join(t) for t in toks]))
        path = os.path.join(dirpath, "bf_%02d.%s" % (f, "txt" if (plain_last and f == n_files - 1) else "gz"))
        data = ("\n".join(lines) + "\n").encode()
        if path.endswith(".gz") and bgzf and f % 2 == 0:
            open(path, "wb").write(_bgzf_bytes(data, member=int(rng.integers(300, 4000))))   # small members: lines straddle them
        elif path.endswith(".gz"):
            with gzip.open(path, "wb", compresslevel=1) as fh:
                fh.write(data)
        else:
            open(path, "wb").write(data)
        paths.append(path)
    return paths


def _bgzf_bytes(data, member=0xff00):
    """`data` as a BGZF file (SAM spec 4.1): gzip members with the BC extra field, raw deflate inside, the empty end marker"""
    import struct
    import zlib
    out = bytearray()
    for at in list(range(0, len(data), member)) + [None]:
        chunk = b"" if at is None else data[at:at + member]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        total = 18 + len(body) + 8
        out += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", total - 1) + body
        out += struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return bytes(out)


def test_pipelined_batchfile_producer_equals_the_plain_loop(tmp_path):
    """basevar_amd/host/batch_producer.hpp (files read and positions parsed as a pipeline of tasks on T threads; BGZF files --
    every other file here, with members so small that lines straddle them -- fetched in segments and inflated by tasks of
    their own) delivers what the position-by-position loop of the reference delivers (src/basetype_caller.cpp:586-611): the
    same positions in order, byte-identical planes and texts on 1 / 2 / 3 / 8 threads; positions nobody covers dropped; the
    run ends at the shortest file; a malformed row ends it with the reference's error AFTER the positions before it."""
    exe = cxx(os.path.join(ROOT, "tests", "cpp", "producer_check.cpp"), str(tmp_path / "pc"), ["-lz"])
    d = tmp_path / "a"; d.mkdir()
    files = _write_batchfiles(str(d), 5, 37, 700, seed=1, plain_last=True)           # several blocks, a plain-text file among them
    out = subprocess.run([exe, ",".join(files)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr
    n_pos = int(out.stdout.split()[1])
    assert 600 < n_pos < 700 and "185 samples" in out.stdout                         # the uncovered positions are gone
    d = tmp_path / "b"; d.mkdir()
    files = _write_batchfiles(str(d), 4, 50, [300, 300, 171, 300], seed=2)           # one file ends early, inside a block
    out = subprocess.run([exe, ",".join(files)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) <= 171
    d = tmp_path / "c"; d.mkdir()
    files = _write_batchfiles(str(d), 3, 40, 400, seed=3)
    import gzip
    lines = gzip.open(files[1], "rb").read().decode().split("\n")
    cols = lines[3 + 250].split("\t")                                               # position 251 of file 1: a two-letter base token
    toks = cols[5].split(" "); toks[7] = "AC"; cols[5] = " ".join(toks); cols[3] = str(int(cols[3]) + 1)
    lines[3 + 250] = "\t".join(cols)
    gzip.open(files[1], "wb").write("\n".join(lines).encode())
    out = subprocess.run([exe, ",".join(files), "Why dose the size of aligned base is not 1? Check: AC"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr
    assert 200 < int(out.stdout.split()[1]) <= 250                                   # the positions before the malformed one


def test_host_formats_harness(tmp_path, restatement):
    exe = cxx(os.path.join(ROOT, "tests", "cpp", "host_formats_check.cpp"), str(tmp_path / "hfc"), ["-ldl"])
    args = [exe, os.path.join(ROOT, "oracle", "liboracle.so")]
    have_ref = oracle.ref_available()
    if have_ref:
        args.append(os.path.join(ROOT, "oracle", "_ref", "libbvref.so"))
    else:
        args.append(os.path.join(ROOT, "oracle", "liboracle.so"))  # placeholder; primitives check is skipped below
    recs = str(tmp_path / "recs.bin")
    if not have_ref:
        pytest.skip("oracle/_ref not available: primitives cannot be pinned here")
    cases = str(tmp_path / "reader_cases.txt")
    out = subprocess.run(args + [recs, cases], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "PRIMITIVES_CHECKED 1" in out.stdout and "FAILS 0" in out.stdout
    assert "FAST_READER_CASES" in out.stdout  # the byte-level batchfile reader against the literal one
    lines = out.stdout.split("\n")
    # headers
    cvg_h = lines[lines.index("CVG_HEADER_BEGIN") + 1:lines.index("CVG_HEADER_END")]
    assert cvg_h == ["##fileformat=CVGv1.0", "##Group information is the depth of A:C:G:T:Indel",
                     "#CHROM\tPOS\tREF\tDepth\tA\tC\tG\tT\tIndels\tFS\tSOR\tStrand_Coverage(REF_FWD,REF_REV,ALT_FWD,ALT_REV)"]
    vcf_h = lines[lines.index("VCF_HEADER_BEGIN") + 1:lines.index("VCF_HEADER_END")]
    assert vcf_h[0] == "##fileformat=VCFv4.2" and vcf_h[-1] == "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2"
    assert "##contig=<ID=chr11,length=135006516,assembly=ref.fa>" in vcf_h and "##reference=file:///abs/ref.fa" in vcf_h
    assert len([l for l in vcf_h if l.startswith("##INFO=")]) == 13 and len([l for l in vcf_h if l.startswith("##FORMAT=")]) == 4
    # records -> lines, re-derived here
    raw = open(recs, "rb").read()
    nrec = sum(1 for l in lines if l.startswith("REC "))
    sites = np.frombuffer(raw[:nrec * 208], dtype=oracle.SITE_DTYPE)
    groups = np.frombuffer(raw[nrec * 208:], dtype=oracle.GROUP_DTYPE).reshape(nrec, 2)
    i = -1
    n_vcf = 0
    for l in lines:
        if l.startswith("REC "):
            i += 1
        elif l.startswith("CVG "):
            c = l[4:].split("\t")
            r = sites[i]
            assert c[3] == str(r["total_depth"]) and c[4:8] == [str(d) for d in r["depth"]]
            assert c[9] == f6(r["cvg_fs"]) and c[10] == f6(r["cvg_sor"]) and c[11] == ",".join(str(x) for x in r["cvg_sb"])
        elif l.startswith("VCF "):
            n_vcf += 1
            v = l[4:].split("\t")
            r = sites[i]
            assert v[4] == ",".join(BASES[b] for b in r["alt"][:r["n_alt"]]) and v[5] == f6(r["qual"])
            assert v[6] == ("." if r["qual"] > 20 else "LowQual") and v[8] == "GT:AB:SO:BP" and len(v) == 9 + 90
            info = dict(kv.split("=") for kv in v[7].split(";"))
            assert list(info)[:12] == ["CM_DP", "CM_AC", "CM_AF", "CM_CAF", "MQRankSum", "ReadPosRankSum", "BaseQRankSum", "QD",
                                       "SOR", "FS", "SB_REF", "SB_ALT"]
            assert info["CM_AF"] == ",".join(g6(x) for x in r["af"][:r["n_alt"]])
            assert info["MQRankSum"] == str(int(r["mq_ranksum"])) and info["QD"] == f6(r["qd"])
            for g, name in enumerate(("BJ", "GD")):
                if groups[i][g]["n_alt"]:
                    assert info[name + "_AF"] == ",".join(g6(x) for x in groups[i][g]["af"][:groups[i][g]["n_alt"]])
                else:
                    assert name + "_AF" not in info
    assert n_vcf >= 5
    # ---- the same sites through the REFERENCE's own per-position caller (text in -> text out): the product's batchfile writer,
    # its formatters and -- through the records -- its reader against the reference's object code, byte for byte
    import ref_caller
    if not ref_caller.available():
        return
    assert ref_caller.cvg_header().split("\n") == cvg_h
    N = 90
    grp = {"BJ": [i for i in range(N) if i % 3 != 0 and i % 2 == 0], "GD": [i for i in range(N) if i % 3 != 0 and i % 2 == 1]}
    maf = restatement.min_af(N, 0.01)
    rows, per_site, cur = [], [], None
    for l in lines:
        if l.startswith("REC "):
            cur = {"rows": [], "cvg": "", "vcf": ""}
            per_site.append(cur)
        elif l.startswith("ROW "):
            cur["rows"].append(l[4:])
        elif l.startswith("CVG "):
            cur["cvg"] = l[4:] + "\n"
        elif l.startswith("VCF "):
            cur["vcf"] = l[4:] + "\n"
    assert len(per_site) == nrec and all(len(s["rows"]) == 3 for s in per_site)
    n_var = 0
    for k, s in enumerate(per_site):
        variant, vcf, cvg = ref_caller.call_position(s["rows"], N, maf, grp)
        assert cvg == s["cvg"], (k, cvg, s["cvg"])
        assert vcf == s["vcf"], (k, vcf[:400], s["vcf"][:400])
        n_var += variant
    assert n_var == n_vcf
    # ---- the byte-level reader's OUTCOME on valid, ragged and damaged rows against the reference's own: taken (a CVG line comes
    # out), skipped (total depth 0: nothing), or an exception -- then with the reference's text.  One deliberate difference: a base
    # character outside ACGTN+- is refused by the product (its slab has no code for it), the reference counts it in the depth.
    raw_cases = open(cases, "rb").read().decode("latin-1").split("\n")
    i = n_cases = n_threw = n_refused = 0
    while i < len(raw_cases) and raw_cases[i].startswith("CASE "):
        head = raw_cases[i].split(" ", 4)
        n_rows, n_smp, kind, what = int(head[1]), int(head[2]), int(head[3]), (head[4] if len(head) > 4 else "").replace("\x01", "\n")
        rows = raw_cases[i + 1:i + 1 + n_rows]
        i += 1 + n_rows
        n_cases += 1
        try:
            _, _, cvg = ref_caller.call_position([r.encode("latin-1").decode("latin-1") for r in rows], n_smp, restatement.min_af(n_smp, 0.01))
            ref_kind, ref_what = (0 if cvg else 1), ""
        except RuntimeError as e:
            ref_kind, ref_what = 2, str(e)
        if kind == 2 and "is outside ACGTN+-" in what:
            n_refused += 1  # (the reference goes on: no CVG line if that was the only read, or an error further down the row)
            continue
        if kind == 0 and ref_kind == 1:
            # taken by the reader, nothing written by the reference: a position whose only reads are indels (its Depth column
            # counts them, the CVG line's depth is over A, C, G, T: basetype_caller.cpp:1241-1249); the product's emitter
            # leaves such a site out the same way (format_cvg_line, checked above on the harness's own sites)
            assert not any(t[:1] in ("A", "C", "G", "T") for r in rows for t in r.split("\t")[5].split(" ")), rows
            continue
        assert kind == ref_kind, (kind, what, ref_kind, ref_what, rows[0][:200])
        if kind == 2:
            n_threw += 1
            assert what == ref_what, (what, ref_what, rows[0][:200])
    assert n_cases > 5000 and n_threw > 1000, (n_cases, n_threw, n_refused)


def make_batchfiles(tmp_path, n_sites=120, n_samples=60, n_files=3, seed=3):
    """Reference-format batchfiles (gzip) for a synthetic region + the per-site token lists."""
    rng = np.random.default_rng(seed)
    per = n_samples // n_files
    ids = ["smp%03d" % i for i in range(n_samples)]
    sites = []
    rows = [[] for _ in range(n_files)]
    for s in range(n_sites):
        ref = BASES[rng.integers(4)]
        alt = BASES[(BASES.index(ref) + 1 + rng.integers(3)) % 4]
        af = [0.0, 0.0, 0.03, 0.3, 0.6][s % 5]
        refcol = ref.lower() if s % 13 == 0 else ("N" if s % 17 == 0 else ref)
        bases, quals, mapqs, ranks, strands = [], [], [], [], []
        for i in range(n_samples):
            if s % 29 != 28 and rng.random() < 0.55:
                b = alt if rng.random() < af else ref
                if rng.random() < 0.02:
                    b = BASES[rng.integers(4)]
                k = rng.integers(50)
                bases.append("+" + b + "G" if k == 0 else ("-" + b if k == 1 else b))
                quals.append(chr(33 + int(rng.integers(4, 42))))
                mapqs.append(60 if rng.random() < 0.8 else int(rng.integers(0, 60)))
                ranks.append(int(rng.integers(1, 151)))
                strands.append("+" if rng.random() < 0.5 else "-")
            else:
                bases.append("N"); quals.append("!"); mapqs.append(0); ranks.append(0); strands.append(".")
        site = {"chrom": "chr17", "pos": 41197764 + s, "ref": refcol, "bases": bases, "quals": quals, "mapqs": mapqs,
                "ranks": ranks, "strands": strands}
        sites.append(site)
        for f in range(n_files):
            sl = slice(f * per, (f + 1) * per)
            cov = sum(1 for t in bases[sl] if t != "N")
            rows[f].append("\t".join([site["chrom"], str(site["pos"]), refcol, str(cov), " ".join(map(str, mapqs[sl])),
                                      " ".join(bases[sl]), " ".join(quals[sl]), " ".join(map(str, ranks[sl])), " ".join(strands[sl])]))
    paths = []
    for f in range(n_files):
        p = str(tmp_path / ("batch_%d.bf.gz" % f))
        with gzip.open(p, "wt") as fh:
            fh.write("##fileformat=BaseVarBatchFile_v1.0\n##SampleIDs=" + ",".join(ids[f * per:(f + 1) * per]) + "\n"
                     "#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\tReadbases\tReadbasesQuality\tReadPositionRank\tStrand\n")
            fh.write("\n".join(rows[f]) + "\n")
        paths.append(p)
    return paths, ids, sites


def reference_caller_lines(paths, n_samples, maf, groups=None):
    """(CVG lines, VCF lines) the REFERENCE's own per-position caller writes for these batchfiles (oracle/_ref/libbvcaller.so:
    `_basevar_caller` compiled where it lies, tests/ref_caller.py); None where that library is not available"""
    import gzip
    import ref_caller
    if not ref_caller.available():
        # (both reference builds come from one recipe, oracle/Makefile, and travel together)
        assert not oracle.ref_available(), "oracle/_ref/libbvref.so is there but libbvcaller.so is not: make -C oracle"
        return None
    per_file = []
    for p in paths:
        opener = gzip.open if open(p, "rb").read(2) == b"\x1f\x8b" else open
        with opener(p, "rt") as fh:
            per_file.append([l.rstrip("\n") for l in fh if not l.startswith("#")])
    assert len(set(len(x) for x in per_file)) == 1
    cvg, vcf = [], []
    for rows in zip(*per_file):
        _, v, c = ref_caller.call_position(list(rows), n_samples, maf, groups)
        cvg += [l for l in c.split("\n") if l]
        vcf += [l for l in v.split("\n") if l]
    return cvg, vcf


def sites_to_slab(sites, n_samples):
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    S = len(sites)
    bs = np.full((S, n_samples), 8, np.uint8); q = np.zeros((S, n_samples), np.uint8)
    mq = np.zeros((S, n_samples), np.uint8); rp = np.zeros((S, n_samples), np.uint16); ref = np.zeros(S, np.uint8)
    for s, site in enumerate(sites):
        ref[s] = code.get(site["ref"][0].upper(), 4)
        for i, t in enumerate(site["bases"]):
            if t[0] == "N":
                continue
            bs[s, i] = 9 if t[0] == "+" else (10 if t[0] == "-" else code[t[0]] | (4 if site["strands"][i] == "-" else 0))
            q[s, i] = ord(site["quals"][i]) - 33
            mq[s, i] = site["mapqs"][i]
            rp[s, i] = site["ranks"][i]
    return {"base_strand": bs, "qual": q, "mapq": mq, "rpr": rp, "ref_base": ref, "n_samples": n_samples}


@pytest.mark.gpu
def test_bv_call_end_to_end(tmp_path, restatement):
    """batchfiles -> bv_call (GPU) -> VCF/CVG text == lines derived from the oracle's records."""
    exe = cxx(os.path.join(ROOT, "basevar_amd", "host", "bv_call.cpp"), str(tmp_path / "bv_call"), ["-lz"])
    n_samples = 60
    paths, ids, sites = make_batchfiles(tmp_path, n_samples=n_samples)
    popfile = str(tmp_path / "groups.info")
    with open(popfile, "w") as fh:
        for i, sid in enumerate(ids):
            if i % 4 != 3:
                fh.write("%s\t%s\n" % (sid, "ZZ" if i % 2 else "AA"))
    vcf, cvg = str(tmp_path / "out.vcf"), str(tmp_path / "out.cvg")
    subprocess.check_call([exe, "--batchfiles", ",".join(paths), "--output-vcf", vcf, "--output-cvg", cvg, "--pop-group",
                           popfile, "--batch-sites", "50", "--contig", "chr17:81195210", "--reference", "hg19.fa"])
    kept = [s for s in sites if any(t != "N" for t in s["bases"])]
    slab = sites_to_slab(kept, n_samples)
    gid = np.full(n_samples, 0xFF, np.uint8)
    for i in range(n_samples):
        if i % 4 != 3:
            gid[i] = 1 if i % 2 else 0   # names sorted: AA -> 0, ZZ -> 1
    slab["group_id"] = gid; slab["n_groups"] = 2
    maf = restatement.min_af(n_samples, 0.01)
    exp, gexp, margins = restatement.run_with_margins(slab, maf)
    got_cvg = [l for l in open(cvg).read().split("\n") if l and not l.startswith("#")]
    got_vcf = [l for l in open(vcf).read().split("\n") if l and not l.startswith("#")]
    exp_cvg = [x for x in (expected_cvg(s, r) for s, r in zip(kept, exp)) if x]
    exp_vcf = [x for x in (expected_vcf(s, r, g, ["AA", "ZZ"]) for s, r, g in zip(kept, exp, gexp)) if x]
    assert len(got_cvg) == len(exp_cvg) and len(got_vcf) == len(exp_vcf) and len(exp_vcf) >= 20

    def same(a, b):
        """identical text, or numerically equal to 1e-6 where the last printed digit may round differently"""
        if a == b:
            return True
        fa, fb = a.replace(";", "\t").replace(",", "\t").replace("=", "\t").replace(":", "\t").split("\t"), \
            b.replace(";", "\t").replace(",", "\t").replace("=", "\t").replace(":", "\t").split("\t")
        if len(fa) != len(fb):
            return False
        for x, y in zip(fa, fb):
            if x == y:
                continue
            try:
                if abs(float(x) - float(y)) > 1e-6 * max(1.0, abs(float(y))) + 1.5e-6:
                    return False
            except ValueError:
                return False
        return True

    tie = margins <= 1e-9
    bad = [(a, b) for a, b in zip(got_cvg, exp_cvg) if not same(a, b)]
    assert not bad, bad[:2]
    exact = sum(1 for a, b in zip(got_vcf, exp_vcf) if a == b)
    bad = [(a[:300], b[:300]) for a, b in zip(got_vcf, exp_vcf) if not same(a, b)]
    assert len(bad) <= int(tie.sum()), bad[:2]
    assert exact >= len(exp_vcf) - int(tie.sum())   # byte-identical but for exact ties (measured: 51 of 51, no tie)
    # the same files through the REFERENCE's own per-position caller: reader, caller, pop-groups and line formatting of the
    # product (batchfiles -> GPU engine -> text) against the reference's object code, line by line
    ref_lines = reference_caller_lines(paths, n_samples, maf, {"AA": [i for i in range(n_samples) if i % 4 != 3 and i % 2 == 0],
                                                               "ZZ": [i for i in range(n_samples) if i % 4 != 3 and i % 2 == 1]})
    if ref_lines is not None:
        assert got_cvg == ref_lines[0]
        assert len(got_vcf) == len(ref_lines[1])
        assert sum(1 for a, b in zip(got_vcf, ref_lines[1]) if a == b) >= len(got_vcf) - int(tie.sum())
        assert all(same(a, b) for a, b in zip(got_vcf, ref_lines[1]))
    hdr = [l for l in open(vcf).read().split("\n") if l.startswith("#")]
    assert hdr[-1].split("\t")[9:] == ids and any(l.startswith("##INFO=<ID=AA_AF") for l in hdr)
    # the block-parallel producer + emitter (`--thread`) write the same bytes (the byte-level reader itself is pinned against the
    # literal restatement of the reference's reader in tests/cpp/host_formats_check.cpp)
    for extra, tag in ((["--thread", "3"], "t3"), (["--thread", "7", "--batch-sites", "13"], "t7")):
        v2, c2 = str(tmp_path / ("out_%s.vcf" % tag)), str(tmp_path / ("out_%s.cvg" % tag))
        subprocess.check_call([exe, "--batchfiles", ",".join(paths), "--output-vcf", v2, "--output-cvg", c2, "--pop-group", popfile,
                               "--batch-sites", "50", "--contig", "chr17:81195210", "--reference", "hg19.fa"] + extra)
        assert open(v2, "rb").read() == open(vcf, "rb").read(), tag
        assert open(c2, "rb").read() == open(cvg, "rb").read(), tag


@pytest.mark.gpu
def test_bv_call_with_more_pop_groups_than_one_round(tmp_path, restatement):
    """40 pop-groups through bv_call (the engine runs pass 2 in rounds of 32 groups; the reference takes any number,
    basetype_caller.cpp:372-410): every <group>_AF field of every VCF line equals the line derived from the oracle's records,
    group names in the reference's (sorted) order, samples without a group left out."""
    exe = cxx(os.path.join(ROOT, "basevar_amd", "host", "bv_call.cpp"), str(tmp_path / "bv_call"), ["-lz"])
    n_samples, n_groups = 240, 40
    paths, ids, sites = make_batchfiles(tmp_path, n_sites=60, n_samples=n_samples, n_files=4, seed=11)
    names = ["pop%02d" % g for g in range(n_groups)]
    popfile = str(tmp_path / "groups.info")
    gid = np.full(n_samples, 0xFF, np.uint8)
    with open(popfile, "w") as fh:
        for i, sid in enumerate(ids):
            if i % 41 != 40:  # a few samples belong to no group
                g = (i * 7) % n_groups
                gid[i] = g
                fh.write("%s\t%s\n" % (sid, names[g]))
    vcf, cvg = str(tmp_path / "out.vcf"), str(tmp_path / "out.cvg")
    subprocess.check_call([exe, "--batchfiles", ",".join(paths), "--output-vcf", vcf, "--output-cvg", cvg, "--pop-group", popfile,
                           "--batch-sites", "25", "--thread", "3"])
    kept = [s for s in sites if any(t != "N" for t in s["bases"])]
    slab = sites_to_slab(kept, n_samples)
    slab["group_id"] = gid; slab["n_groups"] = n_groups
    exp, gexp, margins = restatement.run_with_margins(slab, restatement.min_af(n_samples, 0.01))
    got_vcf = [l for l in open(vcf).read().split("\n") if l and not l.startswith("#")]
    exp_vcf = [x for x in (expected_vcf(s, r, g, names) for s, r, g in zip(kept, exp, gexp)) if x]
    assert len(got_vcf) == len(exp_vcf) >= 15
    n_group_fields = 0
    for a, b in zip(got_vcf, exp_vcf):
        fa, fb = a.split("\t"), b.split("\t")
        assert fa[:5] == fb[:5] and fa[8:] == fb[8:], (a[:200], b[:200])
        ia, ib = fa[7].split(";"), fb[7].split(";")
        assert [x.split("=")[0] for x in ia] == [x.split("=")[0] for x in ib], (fa[7], fb[7])
        for x, y in zip(ia, ib):
            if x.startswith("pop"):
                n_group_fields += 1
                if x != y:  # (a last printed digit may round differently: 1e-6)
                    vx, vy = [float(v) for v in x.split("=")[1].split(",")], [float(v) for v in y.split("=")[1].split(",")]
                    assert len(vx) == len(vy) and all(abs(p - q) <= 1e-6 * max(1.0, abs(q)) + 1.5e-6 for p, q in zip(vx, vy)), (x, y)
    assert n_group_fields >= 10 * len(exp_vcf)  # most groups have an alt read at most variant sites
    hdr = [l for l in open(vcf).read().split("\n") if l.startswith("##INFO=<ID=pop")]
    assert len(hdr) == n_groups
    # ... and against the reference's own per-position caller with the same 40 groups
    ref_lines = reference_caller_lines(paths, n_samples, restatement.min_af(n_samples, 0.01),
                                       {names[g]: [i for i in range(n_samples) if gid[i] == g] for g in range(n_groups)})
    if ref_lines is not None:
        got_cvg = [l for l in open(cvg).read().split("\n") if l and not l.startswith("#")]
        assert got_cvg == ref_lines[0]
        assert len(got_vcf) == len(ref_lines[1])
        n_same = 0
        for a, b in zip(got_vcf, ref_lines[1]):
            if a == b:
                n_same += 1
                continue
            fa, fb = re.split("[\t;,=:]", a), re.split("[\t;,=:]", b)  # (deep sites: a last printed digit may round differently)
            assert len(fa) == len(fb)
            for x, y in zip(fa, fb):
                if x != y:
                    assert abs(float(x) - float(y)) <= 1e-6 * max(1.0, abs(float(y))) + 1.5e-6, (x, y)
        assert n_same >= 0.9 * len(got_vcf), (n_same, len(got_vcf))


@pytest.mark.gpu
def test_bv_call_deep_rows_against_the_reference_caller(tmp_path, restatement):
    """Rows of 2,000 samples (deep sites: the bin-level EM, large QUAL / rank-sum / FS values in the text) in five batchfiles,
    three pop-groups: every CVG line of bv_call equals the reference's own `_basevar_caller` byte for byte, every VCF line
    field for field to 1e-6 (deep sites are within 1e-6 of the reference, not bit-equal: the last printed digit may differ)."""
    exe = cxx(os.path.join(ROOT, "basevar_amd", "host", "bv_call.cpp"), str(tmp_path / "bv_call"), ["-lz"])
    n_samples = 2000
    paths, ids, sites = make_batchfiles(tmp_path, n_sites=45, n_samples=n_samples, n_files=5, seed=23)
    names = ["north", "south", "west"]
    grp = {names[g]: [i for i in range(n_samples) if i % 5 != 4 and i % 3 == g] for g in range(3)}
    popfile = str(tmp_path / "groups.info")
    with open(popfile, "w") as fh:
        for g, idx in grp.items():
            for i in idx:
                fh.write("%s\t%s\n" % (ids[i], g))
    vcf, cvg = str(tmp_path / "out.vcf"), str(tmp_path / "out.cvg")
    subprocess.check_call([exe, "--batchfiles", ",".join(paths), "--output-vcf", vcf, "--output-cvg", cvg, "--pop-group", popfile,
                           "--batch-sites", "16", "--thread", "4"])
    ref_lines = reference_caller_lines(paths, n_samples, restatement.min_af(n_samples, 0.01), grp)
    if ref_lines is None:
        pytest.skip("oracle/_ref/libbvcaller.so not available")
    got_cvg = [l for l in open(cvg).read().split("\n") if l and not l.startswith("#")]
    got_vcf = [l for l in open(vcf).read().split("\n") if l and not l.startswith("#")]
    assert len(got_cvg) == len(ref_lines[0]) >= 40 and len(got_vcf) == len(ref_lines[1]) >= 20
    n_same = 0
    for got, ref in ((got_cvg, ref_lines[0]), (got_vcf, ref_lines[1])):
        for a, b in zip(got, ref):
            if a == b:
                n_same += 1
                continue
            fa, fb = re.split("[\t;,=:]", a), re.split("[\t;,=:]", b)
            assert len(fa) == len(fb), (a[:300], b[:300])
            for x, y in zip(fa, fb):
                if x != y:
                    assert abs(float(x) - float(y)) <= 1e-6 * max(1.0, abs(float(y))) + 1.5e-6, (x, y, a[:200])
    assert n_same >= 0.8 * (len(got_cvg) + len(got_vcf)), (n_same, len(got_cvg), len(got_vcf))


def test_pileup_row_writer_against_the_references_own(tmp_path):
    """SURVEY 8 f2, the writer half: pileup_rows_text (host/pileup.hpp: dense position x sample planes + indel tokens -> batchfile
    rows) against the reference's `__write_record_to_batchfile` (basetype_caller.cpp:1027-1101, compiled where it lies:
    oracle/_ref/libbvcaller.so) on random tiles -- uncovered positions, N calls, insertions, deletions, both strands, ranks to
    65,535: byte for byte.  (The CIGAR walk that fills the planes reads BAM records through htslib and stays unpinned.)"""
    import ref_caller
    oracle.build(with_ref=True)
    if not ref_caller.available():
        pytest.skip("oracle/_ref/libbvcaller.so not available (no reference sources here)")
    exe = str(tmp_path / "pileup_rows_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "pileup_rows_check.cpp"), "-ldl", "-lz", "-o", exe])
    out = subprocess.run([exe, ref_caller.LIB], capture_output=True, text=True)
    assert out.returncode == 0 and "FAILS 0" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_tbi_reader_on_an_index_written_by_htslib():
    """The reference's test data holds one real tabix index (tests/data/chr22.all.sites.vcf.gz.tbi, written by htslib; kept as
    a fixture under tests/golden/).  The independent reader that checks this repo's .tbi writer must read THAT file the way the
    format says: header, one name, bins whose chunks are ordered virtual offsets, the pseudo-bin 37450 (chunk 0 = the span of
    the sequence's records, chunk 1 = record counts), a monotone linear index, and every leaf bin's first chunk at or behind
    its 16 kb window's linear entry -- the invariants test_bgzf_output_and_tabix_index_against_a_linear_scan relies on."""
    from bam_py import read_tbi
    t = read_tbi(os.path.join(ROOT, "tests", "golden", "chr22.all.sites.vcf.gz.tbi"))
    assert t["conf"] == (2, 1, 2, 0, ord("#"), 0)  # htslib's VCF preset: sequence in column 1, position in column 2
    assert t["names"] == ["chr22"] and len(t["refs"]) == 1
    r = t["refs"][0]
    lin = r["linear"]
    assert len(lin) > 1000 and all(a <= b for a, b in zip(lin, lin[1:]))
    pseudo = r["bins"].pop(37450)
    assert len(pseudo) == 2 and pseudo[0][0] < pseudo[0][1] and pseudo[1][0] > 0  # (span of the records), (mapped, unmapped)
    lo, hi = pseudo[0]
    n_leaf = 0
    for b, chunks in r["bins"].items():
        assert 0 <= b < 37449 and chunks
        assert all(beg < end for beg, end in chunks) and all(c[1] <= d[0] for c, d in zip(chunks, chunks[1:]))
        assert lo <= chunks[0][0] and chunks[-1][1] <= hi
        if b >= 4681:  # a 16 kb leaf: window b - 4681 of the linear index starts no later than the bin's first record
            w = b - 4681
            assert w < len(lin) and lin[w] <= chunks[0][0]
            n_leaf += 1
    assert n_leaf > 1000


def test_bgzf_output_and_tabix_index_against_a_linear_scan(tmp_path):
    """`x.gz` outputs (reference: bgzf_write + tbx_index_build, src/basetype_caller.cpp:242-254): the file is a
    sequence of well-formed BGZF blocks ending in the EOF marker (every field, CRC and size checked by an independent reader),
    and the .tbi beside it says, for every data line, where a linear scan finds it: the chunk of its bin covers the line's
    virtual offsets, the linear index of its 16 kb window does not start behind it, the pseudo-bin counts the lines."""
    import bam_py
    exe = cxx(os.path.join(ROOT, "tests", "cpp", "bgzf_tabix_check.cpp"), str(tmp_path / "btc"), ["-lz"])
    gz = str(tmp_path / "t.tsv.gz")
    subprocess.check_call([exe, gz, "3"])
    lines = bam_py.bgzf_lines(gz)
    assert lines[0][2] == b"##fileformat=TESTv1" and lines[1][2].startswith(b"#CHROM")
    recs = [(s, e, l.split(b"\t")) for s, e, l in lines if not l.startswith(b"#")]
    assert len(recs) > 1000 and max(len(c[2]) for _, _, c in recs) == 150000
    # gzip reads the concatenated members as one text
    import gzip
    assert gzip.open(gz, "rb").read() == b"".join(l + b"\n" for _, _, l in lines)
    tbi = bam_py.read_tbi(gz + ".tbi")
    assert tbi["conf"] == (1, 1, 2, 0, ord("#"), 0) and tbi["names"] == ["chr1", "chr2", "chr3"]
    by_ref = {}
    for s, e, c in recs:
        by_ref.setdefault(c[0].decode(), []).append((int(c[1]), s, e))
    for name, ref in zip(tbi["names"], tbi["refs"]):
        mine = by_ref[name]
        meta = ref["bins"].pop(37450)
        assert meta[0] == (mine[0][1], mine[-1][2]) and meta[1] == (len(mine), 0)
        chunks = sorted(c for cs in ref["bins"].values() for c in cs)
        assert all(a[1] <= b[0] for a, b in zip(chunks, chunks[1:]))  # the chunks tile the data lines, in file order
        assert chunks[0][0] == mine[0][1] and chunks[-1][1] == mine[-1][2]
        for pos, s, e in mine:
            beg = pos - 1
            leaf = 4681 + (beg >> 14)
            assert any(cb <= s and e <= ce for cb, ce in ref["bins"].get(leaf, [])), (name, pos)
            assert leaf in bam_py.reg2bins(beg, beg + 1)
            assert ref["linear"][beg >> 14] <= s
        # a window's offset is the first line AT or AFTER the window (empty windows point forward): never behind a line in it
        for w, off in enumerate(ref["linear"]):
            later = [s for pos, s, _ in mine if (pos - 1) >> 14 >= w]
            assert off == (later[0] if later else mine[-1][2])


@pytest.mark.gpu
def test_bam_fixture_end_to_end_counts_of_the_real_binary(tmp_path, restatement):
    """BASELINE configs[0] plumbing: the reference's own test command (tests/data/work.log.sh:1 -- 2 x range.bam,
    ce.fa.gz, CHROMOSOME_I:900-1200, --mapq=10 --min-af=0.05) through bv_pileup -> bv_call (GPU engine).
    The real `basevar basetype` binary printed 5 VCF records and 207 CVG rows for it."""
    data = os.path.join(ROOT, "tests", "golden", "data")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "basevar_amd", "csrc"), "../lib/bv_pileup"], check=True)
    call = cxx(os.path.join(ROOT, "basevar_amd", "host", "bv_call.cpp"), str(tmp_path / "bv_call"), ["-lz"])
    bf = str(tmp_path / "range.bf.gz")
    bam = os.path.join(data, "range.bam")
    subprocess.check_call([os.path.join(ROOT, "basevar_amd", "lib", "bv_pileup"), "-R", os.path.join(data, "ce.fa.gz"), "--regions",
                           "CHROMOSOME_I:900-1200", "--mapq", "10", "-I", bam, "-I", bam, "-o", bf])
    vcf, cvg = str(tmp_path / "vz.vcf"), str(tmp_path / "t.cvg")
    subprocess.check_call([call, "--batchfiles", bf, "--output-vcf", vcf, "--output-cvg", cvg, "--min-af", "0.05"])
    vrec = [l for l in open(vcf).read().split("\n") if l and not l.startswith("#")]
    crow = [l for l in open(cvg).read().split("\n") if l and not l.startswith("#")]
    assert len(crow) == 207 and len(vrec) == 5
    assert open(vcf).read().split("\n#CHROM")[1].split("\n")[0].endswith("FORMAT\tERS225193\tERS225193")
    for l in vrec:
        f = l.split("\t")
        assert f[0] == "CHROMOSOME_I" and 900 <= int(f[1]) <= 1200 and len(f) == 11
    # every line against the line derived from the oracle's record of that position
    sites = []
    for l in gzip.open(bf, "rt").read().splitlines()[3:]:
        c = l.split("\t")
        sites.append({"chrom": c[0], "pos": int(c[1]), "ref": c[2], "mapqs": [int(x) for x in c[4].split(" ")], "bases": c[5].split(" "),
                      "quals": c[6].split(" "), "ranks": [int(x) for x in c[7].split(" ")], "strands": c[8].split(" ")})
    kept = [s_ for s_ in sites if any(t != "N" for t in s_["bases"])]
    pad = 16  # row pitch of the planes
    slab = sites_to_slab(kept, pad)
    slab["n_samples"] = 2
    exp, _, _ = restatement.run_with_margins(slab, restatement.min_af(2, 0.05))
    exp_cvg = [x for x in (expected_cvg(s_, r) for s_, r in zip(kept, exp)) if x]
    exp_vcf = [x for x in (expected_vcf(s_, r, None, []) for s_, r in zip(kept, exp)) if x]
    assert crow == exp_cvg
    assert vrec == exp_vcf
    # BAM inputs handed to bv_call directly (pileup -> engine without the batchfile text): same records
    vcf2, cvg2 = str(tmp_path / "vz2.vcf"), str(tmp_path / "t2.cvg")
    subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200",
                           "--mapq", "10", "--output-vcf", vcf2, "--output-cvg", cvg2, "--min-af", "0.05", "--batch-sites", "64"])
    assert [l for l in open(vcf2).read().split("\n") if l and not l.startswith("##")] == \
        [l for l in open(vcf).read().split("\n") if l and not l.startswith("##")]
    assert open(cvg2).read() == open(cvg).read()
    # outputs named *.gz: BGZF + tabix index (caller.cpp:242-254) -- the same text inside, every CVG row where the index says
    import bam_py
    vz, cz = str(tmp_path / "z.vcf.gz"), str(tmp_path / "z.cvg.gz")
    subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200",
                           "--mapq", "10", "--output-vcf", vz, "--output-cvg", cz, "--min-af", "0.05", "--batch-sites", "64"])
    assert gzip.open(vz, "rb").read() == open(vcf2, "rb").read() and gzip.open(cz, "rb").read() == open(cvg2, "rb").read()
    for path, n_rec in ((vz, 5), (cz, 207)):
        tbi = bam_py.read_tbi(path + ".tbi")
        assert tbi["names"] == ["CHROMOSOME_I"] and tbi["conf"] == (1, 1, 2, 0, ord("#"), 0)
        recs = [(s_, e_, l) for s_, e_, l in bam_py.bgzf_lines(path) if not l.startswith(b"#")]
        assert len(recs) == n_rec and tbi["refs"][0]["bins"][37450][1] == (n_rec, 0)
        for s_, e_, l in recs:
            beg = int(l.split(b"\t")[1]) - 1
            assert any(cb <= s_ and e_ <= ce for cb, ce in tbi["refs"][0]["bins"][4681 + (beg >> 14)])
            assert tbi["refs"][0]["linear"][beg >> 14] <= s_
    # several regions in one call: the records of each, in the order given
    vcf3, cvg3 = str(tmp_path / "vz3.vcf"), str(tmp_path / "t3.cvg")
    subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions",
                           "CHROMOSOME_I:1051-1200,CHROMOSOME_I:900-1050", "--mapq", "10", "--output-vcf", vcf3, "--output-cvg", cvg3,
                           "--min-af", "0.05", "--thread", "2"])
    rows3 = [l for l in open(cvg3).read().split("\n") if l and not l.startswith("#")]
    lo = [l for l in crow if int(l.split("\t")[1]) <= 1050]
    hi = [l for l in crow if int(l.split("\t")[1]) > 1050]
    assert rows3 == hi + lo
    # several engines (here two on the one GPU of the box), small batches finishing out of order: the same files
    vcf4, cvg4 = str(tmp_path / "vz4.vcf"), str(tmp_path / "t4.cvg")
    subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200",
                           "--mapq", "10", "--output-vcf", vcf4, "--output-cvg", cvg4, "--min-af", "0.05", "--batch-sites", "7",
                           "--gpus", "2", "--devices", "0,0", "--thread", "2"])
    assert open(vcf4).read() == open(vcf2).read() and open(cvg4).read() == open(cvg2).read()
    vcf5, cvg5 = str(tmp_path / "vz5.vcf"), str(tmp_path / "t5.cvg")
    subprocess.check_call([call, "--batchfiles", bf, "--output-vcf", vcf5, "--output-cvg", cvg5, "--min-af", "0.05", "--batch-sites", "5",
                           "--devices", "0,0,0"])
    assert open(vcf5).read() == open(vcf).read() and open(cvg5).read() == open(cvg).read()
    # eight engines (the driver's largest node shape, here all on the one GPU), batches of 3 sites: byte-identical to one engine
    vcf8, cvg8 = str(tmp_path / "vz8.vcf"), str(tmp_path / "t8.cvg")
    subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200",
                           "--mapq", "10", "--output-vcf", vcf8, "--output-cvg", cvg8, "--min-af", "0.05", "--batch-sites", "3",
                           "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--thread", "2"])
    assert open(vcf8).read() == open(vcf2).read() and open(cvg8).read() == open(cvg2).read()
    # ... and on two DIFFERENT devices where the box has them (one engine + host thread per GPU, ordered emit: the
    # reference's fan-out and merge, caller.cpp:469-525)
    import torch
    if torch.cuda.device_count() >= 2:
        vcf6, cvg6 = str(tmp_path / "vz6.vcf"), str(tmp_path / "t6.cvg")
        subprocess.check_call([call, "-I", bam, "-I", bam, "-R", os.path.join(data, "ce.fa.gz"), "--regions", "CHROMOSOME_I:900-1200",
                               "--mapq", "10", "--output-vcf", vcf6, "--output-cvg", cvg6, "--min-af", "0.05", "--batch-sites", "7",
                               "--gpus", "2", "--devices", "0,1", "--thread", "2"])
        assert open(vcf6).read() == open(vcf2).read() and open(cvg6).read() == open(cvg2).read()
