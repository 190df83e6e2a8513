"""Independent (pure-Python) BAM reader, BAM writer and pileup used to double-check the C++ host code
(basevar_amd/host/bamio.hpp, pileup.hpp).  Written from the SAM/BAM specification and from the
reference's pileup semantics (src/basetype_caller.cpp:876-1101) -- test infrastructure only."""
import gzip
import struct
import zlib

BASES = "=ACMGRSVTWYHKDBN"
REF_BASES = {1: "A", 2: "C", 4: "G", 8: "T", 15: "N"}  # everything else prints as ' ' (src/bam_record.h:28-31)
CIGAR_OPS = "MIDNSHP=X"


def read_bam(path):
    """-> (header_text, [(name, length)], [record dict])"""
    data = gzip.open(path, "rb").read()
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, o); o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, o); o += 4
        name = data[o:o + l_name - 1].decode(); o += l_name
        l_ref, = struct.unpack_from("<i", data, o); o += 4
        refs.append((name, l_ref))
    recs = []
    while o < len(data):
        bs, = struct.unpack_from("<i", data, o); o += 4
        tid, pos, l_rn, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", data, o)
        p = o + 32 + l_rn
        cigar = [(c & 15, c >> 4) for c in struct.unpack_from("<%dI" % n_cig, data, p)]
        p += 4 * n_cig
        seq = "".join(REF_BASES.get((data[p + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15, " ") for i in range(l_seq))
        p += (l_seq + 1) // 2
        qual = list(data[p:p + l_seq])
        recs.append(dict(tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual))
        o += bs
    return text, refs, recs


def write_bam(path, refs, records, header_text=None, block_payload=60000):
    """records: dicts with tid,pos,mapq,flag,cigar [(op,len)],seq (ACGTN string),qual [ints],name"""
    if header_text is None:
        header_text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + "@RG\tID:x\tSM:synth\n"
    raw = bytearray(b"BAM\x01")
    t = header_text.encode()
    raw += struct.pack("<i", len(t)) + t + struct.pack("<i", len(refs))
    for name, ln in refs:
        raw += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", ln)
    code = {c: i for i, c in enumerate(BASES)}
    for r in records:
        name = r.get("name", "r").encode() + b"\0"
        seq = r["seq"]
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i >> 1] |= code[c] << (4 if i % 2 == 0 else 0)
        cig = b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in r["cigar"])
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], 4680, len(r["cigar"]), r["flag"], len(seq),
                           -1, -1, 0) + name + cig + bytes(packed) + bytes(r["qual"])
        raw += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        for s in range(0, len(raw), block_payload):
            f.write(_bgzf_block(bytes(raw[s:s + block_payload])))
        f.write(_bgzf_block(b""))


def _bgzf_block(payload):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(payload) + c.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + comp +
            struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


def read_fasta(path, ref_id):
    seq, on = [], False
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    for line in op(path, "rt"):
        if line.startswith(">"):
            if on:
                break
            on = line[1:].split()[0] == ref_id
        elif on:
            seq.append(line.strip())
    return "".join(seq)


def sample_name(text):
    for line in text.split("\n"):
        if line.startswith("@RG"):
            for f in line.split("\t")[1:]:
                if f.startswith("SM:"):
                    return f[3:]
    raise ValueError("no SM")


def end_pos(rec):
    rlen = sum(ln for op, ln in rec["cigar"] if CIGAR_OPS[op] in "MDN=X")
    return rec["pos"] + (rlen or 1)


def pileup_sample(recs, tid, fa, reg_start, reg_end, mapq_thd):
    """first-read-wins cells of one sample in [reg_start, reg_end] (1-based): {pos: (mapq, token, qualchar, rank, strand)}"""
    cells = {}
    for r in recs:
        if r["tid"] != tid:
            continue
        # the iterator: reads overlapping the padded region, file order
        lo, hi = max(reg_start - 200, 1) - 1, reg_end + 200
        if r["pos"] >= hi or end_pos(r) <= lo:
            continue
        if r["mapq"] < mapq_thd:
            continue
        if r["flag"] & 4:  # unmapped: start -1 / end -1 never overlaps
            continue
        if r["flag"] & (1024 | 512):
            continue
        start1, end1 = r["pos"] + 1, end_pos(r)
        if reg_start > end1:
            continue
        if reg_end < start1:
            break
        strand = "-" if r["flag"] & 16 else "+"
        meanq = chr(int(sum(r["qual"]) / len(r["qual"])) + 33) if r["seq"] else chr(32)
        rpos, qpos = r["pos"], 0  # 0-based
        stop = False
        for op, ln in r["cigar"]:
            o = CIGAR_OPS[op]
            if o in "M=X":
                for k in range(ln):
                    p1 = rpos + k + 1
                    if p1 > reg_end:
                        stop = True
                        break
                    if p1 >= reg_start and p1 not in cells:
                        cells[p1] = (r["mapq"], r["seq"][qpos + k], chr(r["qual"][qpos + k] + 33), qpos + k + 1, strand)
                if stop:
                    break
                rpos += ln; qpos += ln
            elif o == "I":
                p1 = rpos + 1  # un-anchored position decides about the region
                if p1 > reg_end:
                    break
                if p1 >= reg_start and rpos not in cells:
                    cells[rpos] = (r["mapq"], "+" + fa[rpos - 1] + r["seq"][qpos:qpos + ln], meanq, qpos + 1, strand)
                qpos += ln
            elif o == "D":
                p1 = rpos + 1
                if p1 > reg_end:
                    break
                if p1 >= reg_start and rpos not in cells:
                    cells[rpos] = (r["mapq"], "-" + fa[rpos - 1] + fa[rpos:rpos + ln], meanq, qpos + 1, strand)
                rpos += ln
            elif o == "N":
                if rpos + 1 > reg_end:
                    break
                rpos += ln
            elif o in "SP":
                if rpos + 1 > reg_end:
                    break
                qpos += ln
            # H: nothing
    return cells


def batchfile_text(bam_paths, fasta, ref_id, reg_start, reg_end, mapq_thd, step=500000):
    fa = read_fasta(fasta, ref_id)
    loaded = [read_bam(p) for p in bam_paths]
    ids = [sample_name(t) for t, _, _ in loaded]
    out = ["##fileformat=BaseVarBatchFile_v1.0\n##SampleIDs=%s\n#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\t"
           "Readbases\tReadbasesQuality\tReadPositionRank\tStrand\n" % ",".join(ids)]
    for sb in range(reg_start, reg_end + 1, step):
        se = min(sb + step - 1, reg_end)
        per = []
        for text, refs, recs in loaded:
            tid = [n for n, _ in refs].index(ref_id)
            per.append(pileup_sample(recs, tid, fa, sb, se, mapq_thd))
        for pos in range(sb, se + 1):
            cells = [c.get(pos) for c in per]
            depth = sum(c is not None for c in cells)
            cols = [[str(c[0]) if c else "0" for c in cells], [c[1] if c else "N" for c in cells],
                    [c[2] if c else "!" for c in cells], [str(c[3]) if c else "0" for c in cells],
                    [c[4] if c else "." for c in cells]]
            out.append("%s\t%d\t%s\t%d\t%s\n" % (ref_id, pos, fa[pos - 1], depth, "\t".join(" ".join(x) for x in cols)))
    return "".join(out)


# ---------------------------------------------------------------- BGZF blocks and the tabix index, read independently
def bgzf_blocks(path):
    """-> [(file offset, compressed size, payload bytes)] of every BGZF block, checked field by field (SAM spec 4.1);
    the last block must be the empty end-of-file marker."""
    data = open(path, "rb").read()
    out, o = [], 0
    while o < len(data):
        assert data[o:o + 4] == b"\x1f\x8b\x08\x04", "gzip member with FEXTRA at %d" % o
        xlen, = struct.unpack_from("<H", data, o + 10)
        assert xlen == 6 and data[o + 12:o + 14] == b"BC" and struct.unpack_from("<H", data, o + 14)[0] == 2
        bsize, = struct.unpack_from("<H", data, o + 16)
        total = bsize + 1
        comp = data[o + 18:o + total - 8]
        crc, isize = struct.unpack_from("<II", data, o + total - 8)
        payload = zlib.decompress(comp, -15)
        assert len(payload) == isize and (zlib.crc32(payload) & 0xffffffff) == crc and isize <= 65536
        out.append((o, total, payload))
        o += total
    assert out and out[-1][2] == b"" and out[-1][1] == 28, "the file ends with the 28-byte EOF block"
    return out


def bgzf_lines(path):
    """-> [(virtual offset of the line's first byte, virtual offset behind its newline, line bytes without the newline)]"""
    blocks = bgzf_blocks(path)
    lines, cur, start = [], b"", None
    for bi, (off, _, payload) in enumerate(blocks):
        for i in range(len(payload)):
            if start is None:
                start = (off << 16) | i
            if payload[i] == 10:
                # the offset behind the newline: inside this block -- or, when the newline filled the block to the writer's block
                # size (0xff00: it was flushed at once), the start of the next block; both name the same byte
                end = ((off << 16) | (i + 1)) if (i + 1 < len(payload) or len(payload) != 0xff00) else (blocks[bi + 1][0] << 16)
                lines.append((start, end, cur))
                cur, start = b"", None
            else:
                cur += payload[i:i + 1]
    assert cur == b""
    return lines


def read_tbi(path):
    """-> dict(conf, names, refs=[dict(bins={bin: [(beg, end)]}, linear=[...])])"""
    data = b"".join(p for _, _, p in bgzf_blocks(path))
    assert data[:4] == b"TBI\x01"
    n_ref, preset, sc, bc, ec, meta, skip, l_nm = struct.unpack_from("<8i", data, 4)
    o = 36
    names = data[o:o + l_nm].split(b"\0")[:-1]
    assert len(names) == n_ref
    o += l_nm
    refs = []
    for _ in range(n_ref):
        n_bin, = struct.unpack_from("<i", data, o); o += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", data, o); o += 8
            bins[b] = [struct.unpack_from("<QQ", data, o + 16 * k) for k in range(n_chunk)]
            o += 16 * n_chunk
        n_intv, = struct.unpack_from("<i", data, o); o += 4
        linear = list(struct.unpack_from("<%dQ" % n_intv, data, o)); o += 8 * n_intv
        refs.append(dict(bins=bins, linear=linear))
    assert o == len(data) or o + 8 == len(data)
    return dict(conf=(preset, sc, bc, ec, meta, skip), names=[n.decode() for n in names], refs=refs)


def reg2bins(beg, end):
    """the bins that may hold records overlapping [beg, end) (the tabix / BAM binning scheme)"""
    end -= 1
    bins = [0]
    for shift, first in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(first + (beg >> shift), first + (end >> shift) + 1))
    return bins
