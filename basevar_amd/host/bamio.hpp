// bamio.hpp -- reading BAM alignments for the pileup, without htslib:
// BGZF blocks through zlib, the BAM binary layout, and the BAI binning index for region queries
// (SAM/BAM specification, sections 4.1-4.2 and 5.1-5.2).  It provides what the reference takes from
// ngslib::Bam / BamRecord / BamHeader (src/bam.h, src/bam_record.h, src/bam_header.cpp), i.e. from
// htslib's sam_itr_querys / sam_itr_next / bam_endpos / sam_hdr_find_tag_pos:
//   open + header, first @RG SM tag, "reads overlapping a region, in file order", and the record
//   fields the pileup uses.  CRAM is not read (it needs its own reference-based codecs).
//
// PARITY STATUS: unpinned against htslib (not buildable under this round's rules); checked against an
// independent Python reader and against indexed-vs-linear scans (tests/test_pileup_cpu.py).
#pragma once

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace bvamd {

// ------------------------------------------------------------------------------------------ BGZF
// One gzip member per block (<= 64 KiB of payload), its total size in the 'BC' extra subfield.
// A virtual offset is (file offset of the block << 16) | offset inside the inflated block.
class BgzfReader {
public:
    explicit BgzfReader(const std::string &path) : path_(path) {
        f_ = std::fopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("[ERROR] cannot open " + path);
    }
    ~BgzfReader() { if (f_) std::fclose(f_); }
    BgzfReader(const BgzfReader &) = delete;
    BgzfReader &operator=(const BgzfReader &) = delete;

    uint64_t tell() const { return (block_addr_ << 16) | (uint64_t)pos_; }
    void seek(uint64_t voffset) {
        const uint64_t addr = voffset >> 16;
        if (addr != block_addr_ || !have_block_) load_block(addr);
        pos_ = (size_t)(voffset & 0xFFFFu);
        if (pos_ > data_.size()) throw std::runtime_error("[ERROR] bad virtual offset in " + path_);
    }
    // false at end of file when nothing was read; throws on a truncated item
    bool read(void *dst, size_t n) {
        uint8_t *d = static_cast<uint8_t *>(dst);
        size_t got = 0;
        while (got < n) {
            if (!have_block_ || pos_ == data_.size()) {
                if (!load_block(have_block_ ? next_addr_ : 0)) {
                    if (got == 0) return false;
                    throw std::runtime_error("[ERROR] truncated file " + path_);
                }
                continue;
            }
            const size_t k = std::min(n - got, data_.size() - pos_);
            std::memcpy(d + got, data_.data() + pos_, k);
            got += k;
            pos_ += k;
        }
        return true;
    }

private:
    bool load_block(uint64_t addr) {
        for (;;) {  // empty blocks (e.g. the EOF marker) are skipped
            if (std::fseek(f_, (long)addr, SEEK_SET) != 0) throw std::runtime_error("[ERROR] seek failed in " + path_);
            uint8_t h[18];
            const size_t r = std::fread(h, 1, sizeof h, f_);
            if (r == 0) return false;
            if (r >= 4 && std::memcmp(h, "CRAM", 4) == 0)
                throw std::runtime_error("[ERROR] " + path_ + " is a CRAM file: only BAM is read here (samtools view -b converts)");
            if (r != sizeof h || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
                throw std::runtime_error("[ERROR] not a BGZF file: " + path_);
            const unsigned xlen = h[10] | (h[11] << 8);
            // the BC subfield is the first (and normally only) one; be tolerant and search it
            std::vector<uint8_t> extra(xlen);
            std::memcpy(extra.data(), h + 12, std::min<size_t>(6, xlen));
            if (xlen > 6 && std::fread(extra.data() + 6, 1, xlen - 6, f_) != xlen - 6)
                throw std::runtime_error("[ERROR] truncated BGZF header in " + path_);
            int bsize = -1;
            for (size_t i = 0; i + 4 <= xlen;) {
                const unsigned slen = extra[i + 2] | (extra[i + 3] << 8);
                if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = extra[i + 4] | (extra[i + 5] << 8);
                i += 4 + slen;
            }
            if (bsize < 0) throw std::runtime_error("[ERROR] BGZF block without BC field in " + path_);
            const size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
            std::vector<uint8_t> comp(clen + 8);
            if (std::fread(comp.data(), 1, clen + 8, f_) != clen + 8) throw std::runtime_error("[ERROR] truncated BGZF block in " + path_);
            const uint32_t isize = comp[clen + 4] | (comp[clen + 5] << 8) | (comp[clen + 6] << 16) | ((uint32_t)comp[clen + 7] << 24);
            data_.resize(isize);
            if (isize) {
                z_stream zs;
                std::memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("[ERROR] zlib init failed");
                zs.next_in = comp.data(); zs.avail_in = (uInt)clen;
                zs.next_out = data_.data(); zs.avail_out = isize;
                const int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END || zs.total_out != isize) throw std::runtime_error("[ERROR] corrupt BGZF block in " + path_);
            }
            block_addr_ = addr;
            next_addr_ = addr + (uint64_t)bsize + 1;
            pos_ = 0;
            have_block_ = true;
            if (isize) return true;
            addr = next_addr_;
        }
    }
    std::string path_;
    std::FILE *f_ = nullptr;
    std::vector<uint8_t> data_;
    uint64_t block_addr_ = 0, next_addr_ = 0;
    size_t pos_ = 0;
    bool have_block_ = false;
};

// ------------------------------------------------------------------------------------------- BAM
enum { BAM_CMATCH = 0, BAM_CINS = 1, BAM_CDEL = 2, BAM_CREF_SKIP = 3, BAM_CSOFT_CLIP = 4, BAM_CHARD_CLIP = 5,
       BAM_CPAD = 6, BAM_CEQUAL = 7, BAM_CDIFF = 8 };
enum { BAM_FUNMAP = 4, BAM_FREVERSE = 16, BAM_FQCFAIL = 512, BAM_FDUP = 1024 };

// the members of ngslib::BamRecord the pileup reads (src/bam_record.h:150-240, src/bam_record.cpp:217-345)
struct BamAlignment {
    int32_t tid = -1, pos = -1;  // 0-based leftmost coordinate
    uint8_t mapq_ = 0;
    uint16_t flag = 0;
    std::vector<uint32_t> cigar;  // len << 4 | op
    std::string seq;              // _BASES[] letters, src/bam_record.h:28-31
    std::vector<uint8_t> qual;    // phred, no +33

    bool is_mapped() const { return !(flag & BAM_FUNMAP); }
    bool is_duplicate() const { return is_mapped() && (flag & BAM_FDUP); }
    bool is_qc_fail() const { return is_mapped() && (flag & BAM_FQCFAIL); }
    int mapq() const { return mapq_; }
    char map_strand() const { return is_mapped() ? ((flag & BAM_FREVERSE) ? '-' : '+') : '*'; }
    int64_t map_ref_start_pos() const { return is_mapped() ? pos : -1; }
    // bam_endpos: first base after the alignment, 0-based (== last aligned base, 1-based); pos + 1 when
    // the CIGAR consumes no reference
    int64_t end_pos() const {
        int64_t rlen = 0;
        for (uint32_t c : cigar) {
            const int op = c & 15;
            if (op == BAM_CMATCH || op == BAM_CDEL || op == BAM_CREF_SKIP || op == BAM_CEQUAL || op == BAM_CDIFF) rlen += c >> 4;
        }
        return pos + (rlen ? rlen : 1);
    }
    int64_t map_ref_end_pos() const { return is_mapped() ? end_pos() : -1; }
    double mean_qqual() const {  // src/bam_record.cpp:332-343
        if (!is_mapped() || seq.empty()) return -1;
        double total = 0;
        for (uint8_t q : qual) total += q;
        return total / (double)seq.size();
    }
};

struct BamRef {
    std::string name;
    uint32_t length;
};

// BAI: per reference, bins -> chunks of virtual offsets, plus the 16 kb linear index
struct BaiIndex {
    struct Ref {
        std::vector<std::pair<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>>> bins;
        std::vector<uint64_t> linear;
    };
    std::vector<Ref> refs;
    bool loaded = false;

    static uint32_t rd32(std::FILE *f) { uint8_t b[4]; if (std::fread(b, 1, 4, f) != 4) throw std::runtime_error("[ERROR] truncated BAI"); return b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24); }
    static uint64_t rd64(std::FILE *f) { uint64_t lo = rd32(f), hi = rd32(f); return lo | (hi << 32); }
    bool load(const std::string &path) {
        std::FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        try {
            char magic[4];
            if (std::fread(magic, 1, 4, f) != 4 || std::memcmp(magic, "BAI\1", 4) != 0) throw std::runtime_error("[ERROR] not a BAI file: " + path);
            const uint32_t n_ref = rd32(f);
            refs.resize(n_ref);
            for (auto &r : refs) {
                const uint32_t n_bin = rd32(f);
                r.bins.resize(n_bin);
                for (auto &b : r.bins) {
                    b.first = rd32(f);
                    const uint32_t n_chunk = rd32(f);
                    b.second.resize(n_chunk);
                    for (auto &c : b.second) { c.first = rd64(f); c.second = rd64(f); }
                }
                const uint32_t n_intv = rd32(f);
                r.linear.resize(n_intv);
                for (auto &o : r.linear) o = rd64(f);
            }
        } catch (...) { std::fclose(f); throw; }
        std::fclose(f);
        loaded = true;
        return true;
    }
    // chunks that may hold alignments overlapping [beg, end) (0-based), merged and sorted (SAM spec 5.3)
    std::vector<std::pair<uint64_t, uint64_t>> query(int tid, int64_t beg, int64_t end) const {
        std::vector<std::pair<uint64_t, uint64_t>> out;
        if (tid < 0 || (size_t)tid >= refs.size()) return out;
        const Ref &r = refs[tid];
        if (end > (1ll << 29)) end = 1ll << 29;
        if (beg < 0) beg = 0;
        if (beg >= end) return out;
        std::vector<uint32_t> want;
        want.push_back(0);
        --end;
        for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) want.push_back((uint32_t)k);
        for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) want.push_back((uint32_t)k);
        for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) want.push_back((uint32_t)k);
        for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) want.push_back((uint32_t)k);
        for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) want.push_back((uint32_t)k);
        uint64_t min_off = 0;
        const size_t win = (size_t)(beg >> 14);
        if (!r.linear.empty()) min_off = r.linear[std::min(win, r.linear.size() - 1)];
        for (const auto &b : r.bins) {
            if (b.first == 37450u) continue;  // metadata pseudo-bin
            if (std::find(want.begin(), want.end(), b.first) == want.end()) continue;
            for (const auto &c : b.second)
                if (c.second > min_off) out.push_back(c);
        }
        std::sort(out.begin(), out.end());
        std::vector<std::pair<uint64_t, uint64_t>> merged;
        for (const auto &c : out) {
            if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second);
            else merged.push_back(c);
        }
        return merged;
    }
};

// ngslib::Bam: open, header, fetch(region) + next()
class BamFile {
public:
    explicit BamFile(const std::string &path, bool use_index = true) : path_(path), bg_(path) {
        char magic[4];
        if (!bg_.read(magic, 4) || std::memcmp(magic, "BAM\1", 4) != 0) throw std::runtime_error("[ERROR] not a BAM file: " + path);
        const uint32_t l_text = rd32();
        text_.resize(l_text);
        if (l_text && !bg_.read(&text_[0], l_text)) throw std::runtime_error("[ERROR] truncated BAM header in " + path);
        const uint32_t n_ref = rd32();
        refs_.resize(n_ref);
        for (auto &r : refs_) {
            const uint32_t l_name = rd32();
            std::string nm(l_name, '\0');
            if (l_name && !bg_.read(&nm[0], l_name)) throw std::runtime_error("[ERROR] truncated BAM header in " + path);
            if (!nm.empty() && nm.back() == '\0') nm.pop_back();
            r.name = nm;
            r.length = rd32();
        }
        first_record_ = bg_.tell();
        if (use_index && !index_.load(path + ".bai")) {
            const size_t dot = path.rfind('.');
            if (dot != std::string::npos) index_.load(path.substr(0, dot) + ".bai");
        }
    }
    const std::string &header_text() const { return text_; }
    const std::vector<BamRef> &refs() const { return refs_; }
    bool has_index() const { return index_.loaded; }
    int tid_of(const std::string &name) const {
        for (size_t i = 0; i < refs_.size(); ++i)
            if (refs_[i].name == name) return (int)i;
        return -1;
    }
    // BamHeader::get_sample_name (src/bam_header.cpp:62-83): SM of the first @RG line that has one
    std::string sample_name() const {
        size_t p = 0;
        while (p < text_.size()) {
            size_t e = text_.find('\n', p);
            if (e == std::string::npos) e = text_.size();
            const std::string line = text_.substr(p, e - p);
            if (line.compare(0, 3, "@RG") == 0) {
                size_t q = 0;
                while ((q = line.find('\t', q)) != std::string::npos) {
                    ++q;
                    if (line.compare(q, 3, "SM:") == 0) {
                        size_t t = line.find('\t', q);
                        return line.substr(q + 3, (t == std::string::npos ? line.size() : t) - q - 3);
                    }
                }
            }
            p = e + 1;
        }
        throw std::runtime_error("[bam_header.cpp::BamHeader:get_sample_name] Bam file format error: "
                                 "missing `SM` tag in `@RG` field in BAM/CRAM/SAM header.");
    }
    // Bam::fetch(region): afterwards next() yields the alignments overlapping [beg, end) of `tid`
    // (0-based, half open) in file order.  False if the reference is not in this file.
    bool fetch(int tid, int64_t beg, int64_t end) {
        if (tid < 0 || (size_t)tid >= refs_.size()) return false;
        q_tid_ = tid; q_beg_ = beg; q_end_ = end;
        chunks_.clear();
        chunk_i_ = 0;
        done_ = false;
        if (index_.loaded) {
            chunks_ = index_.query(tid, beg, end);
            if (chunks_.empty()) { done_ = true; return true; }
            bg_.seek(chunks_[0].first);
        } else {
            bg_.seek(first_record_);
        }
        return true;
    }
    // >= 0 on a record, -1 at the end of the query (the reference's `while (bf.next(al) >= 0)`)
    int next(BamAlignment &al) {
        while (!done_) {
            if (index_.loaded) {
                while (chunk_i_ < chunks_.size() && bg_.tell() >= chunks_[chunk_i_].second) {
                    ++chunk_i_;
                    if (chunk_i_ < chunks_.size() && bg_.tell() < chunks_[chunk_i_].first) bg_.seek(chunks_[chunk_i_].first);
                }
                if (chunk_i_ >= chunks_.size()) break;
            }
            if (!read_record(al)) break;
            if (al.tid != q_tid_) {
                if (al.tid > q_tid_ || al.tid < 0) break;  // sorted file: past the reference
                continue;
            }
            if (al.pos >= q_end_) break;                    // sorted by position
            if (al.end_pos() <= q_beg_) continue;
            return 0;
        }
        done_ = true;
        return -1;
    }

private:
    uint32_t rd32() {
        uint8_t b[4];
        if (!bg_.read(b, 4)) throw std::runtime_error("[ERROR] truncated BAM file " + path_);
        return b[0] | (b[1] << 8) | (b[2] << 16) | ((uint32_t)b[3] << 24);
    }
    bool read_record(BamAlignment &al) {
        uint8_t b4[4];
        if (!bg_.read(b4, 4)) return false;
        const uint32_t block_size = b4[0] | (b4[1] << 8) | (b4[2] << 16) | ((uint32_t)b4[3] << 24);
        if (block_size < 32) throw std::runtime_error("[ERROR] corrupt BAM record in " + path_);
        buf_.resize(block_size);
        if (!bg_.read(buf_.data(), block_size)) throw std::runtime_error("[ERROR] truncated BAM record in " + path_);
        const uint8_t *p = buf_.data();
        auto u32 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); };
        al.tid = (int32_t)u32(0);
        al.pos = (int32_t)u32(4);
        const unsigned l_read_name = p[8];
        al.mapq_ = p[9];
        const unsigned n_cigar = p[12] | (p[13] << 8);
        al.flag = (uint16_t)(p[14] | (p[15] << 8));
        const uint32_t l_seq = u32(16);
        size_t o = 32 + l_read_name;
        if (o + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq > block_size) throw std::runtime_error("[ERROR] corrupt BAM record in " + path_);
        al.cigar.resize(n_cigar);
        for (unsigned i = 0; i < n_cigar; ++i) al.cigar[i] = u32(o + 4 * i);
        o += 4ull * n_cigar;
        static const char BASES[16] = {' ', 'A', 'C', ' ', 'G', ' ', ' ', ' ', 'T', ' ', ' ', ' ', ' ', ' ', ' ', 'N'};
        al.seq.resize(l_seq);
        for (uint32_t i = 0; i < l_seq; ++i) al.seq[i] = BASES[(p[o + (i >> 1)] >> ((~i & 1) << 2)) & 15];
        o += (l_seq + 1) / 2;
        al.qual.assign(p + o, p + o + l_seq);
        return true;
    }
    std::string path_;
    BgzfReader bg_;
    std::string text_;
    std::vector<BamRef> refs_;
    BaiIndex index_;
    uint64_t first_record_ = 0;
    std::vector<uint8_t> buf_;
    int q_tid_ = -1;
    int64_t q_beg_ = 0, q_end_ = 0;
    std::vector<std::pair<uint64_t, uint64_t>> chunks_;
    size_t chunk_i_ = 0;
    bool done_ = true;
};

}  // namespace bvamd
