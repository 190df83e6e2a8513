// bgzf_tabix.hpp -- compressed, indexed text output: a BGZF writer and a tabix (.tbi) index builder.
//
// The reference writes `x.vcf.gz` / `x.cvg.gz` through htslib's bgzf_write and indexes them with
// tbx_index_build(fn, 0, {preset 1, seq col 1, beg col 2, end col 0, meta '#', skip 0}) (src/basetype_caller.cpp:242-254,
// src/basetype_utils.cpp:95-96).  No htslib here: this is an own writer of the two published formats (SAM/BAM spec section 4.1
// "The BGZF compression format"; the tabix index layout of the tabix paper / htslib's tbx.c header), zlib only.
//
//   BgzfWriter    gzip members of at most 0xff00 bytes of payload, each with the BC extra field that holds its compressed size
//                 (so that a reader can seek to a block), raw deflate inside, CRC32 + ISIZE behind; the 28-byte empty block as
//                 end-of-file marker.  tell() is the VIRTUAL offset of the next byte: (file offset of its block) << 16 | (offset
//                 inside the block's payload).
//   TabixIndex    one call per data line (sequence name, 1-based position, virtual offsets before and after the line):
//                 the binning index (UCSC bins over 2^29 bases, 16 kb leaves: runs of consecutive lines in one bin become a chunk),
//                 the linear index (per 16 kb window the smallest offset of a line in it; empty windows take the next one's), the
//                 per-sequence pseudo-bin 37450 (range of offsets, line count) and the header with the column configuration and
//                 the sequence names -- itself written as a BGZF file.
//
// PARITY STATUS: both files follow the published formats and are read back by the independent Python readers of the tests
// (tests/bam_py.py; tests/test_host_formats.py checks every index offset against a linear scan); byte-level equality with
// htslib's output is unpinned (no htslib build here) and not aimed at: htslib merges small bins into their parents when it
// finishes an index, which changes the file, not what a query returns.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace bvamd {

class BgzfWriter {
public:
    static constexpr size_t kBlock = 0xff00;  // payload bytes per block (what htslib uses: the compressed block stays under 64 KiB)
    BgzfWriter() = default;
    BgzfWriter(const BgzfWriter &) = delete;
    BgzfWriter &operator=(const BgzfWriter &) = delete;
    ~BgzfWriter() { try { close(); } catch (...) { } }

    void open(const std::string &path, int level = Z_DEFAULT_COMPRESSION) {
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) throw std::runtime_error("[ERROR] " + path + " open failure.");
        path_ = path; level_ = level; buf_.clear(); buf_.reserve(kBlock); block_off_ = 0;
    }
    bool is_open() const { return f_ != nullptr; }
    // the virtual offset of the next byte written
    uint64_t tell() const { return (block_off_ << 16) | (uint64_t)buf_.size(); }
    void write(const char *p, size_t n) {
        while (n) {
            const size_t take = n < kBlock - buf_.size() ? n : kBlock - buf_.size();
            buf_.insert(buf_.end(), p, p + take);
            p += take; n -= take;
            if (buf_.size() == kBlock) flush_block();
        }
    }
    void write(const std::string &s) { write(s.data(), s.size()); }
    // end the current block here (a reader can then start at tell() without inflating what came before)
    void flush() { if (!buf_.empty()) flush_block(); }
    void close() {
        if (!f_) return;
        flush();
        static const unsigned char eof_marker[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool ok = std::fwrite(eof_marker, 1, sizeof eof_marker, f_) == sizeof eof_marker;
        const bool closed = std::fclose(f_) == 0;
        f_ = nullptr;
        if (!ok || !closed) throw std::runtime_error("[ERROR] write failure on " + path_);
    }

private:
    void flush_block() {
        unsigned char out[0x10000];
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level_, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("[ERROR] deflateInit2 failed");
        zs.next_in = reinterpret_cast<Bytef *>(buf_.data());
        zs.avail_in = (uInt)buf_.size();
        zs.next_out = out + 18;
        zs.avail_out = sizeof out - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw std::runtime_error("[ERROR] a BGZF block did not fit 64 KiB compressed");  // (0xff00 bytes always do)
        const size_t total = 18 + clen + 8;
        static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(out, head, 16);
        out[16] = (unsigned char)((total - 1) & 0xff); out[17] = (unsigned char)((total - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(buf_.data()), (uInt)buf_.size());
        const uint32_t isize = (uint32_t)buf_.size();
        for (int i = 0; i < 4; ++i) { out[18 + clen + i] = (unsigned char)(crc >> (8 * i)); out[22 + clen + i] = (unsigned char)(isize >> (8 * i)); }
        if (std::fwrite(out, 1, total, f_) != total) throw std::runtime_error("[ERROR] write failure on " + path_);
        block_off_ += total;
        buf_.clear();
    }
    FILE *f_ = nullptr;
    std::string path_;
    int level_ = Z_DEFAULT_COMPRESSION;
    std::vector<char> buf_;
    uint64_t block_off_ = 0;
};

// the column configuration written into the index: the reference's {1, 1, 2, 0, '#', 0} (caller.cpp:242).  Preset 1 is htslib's
// SAM preset: with it tbx.c takes a record's END from column 6 read as a CIGAR string; the reference's column 6 is a number
// (QUAL in the VCF, a depth in the CVG file), which that parser turns into a length of 1 -- every line covers the one base at
// its position, which is what add_line() below indexes.
struct TabixConf { int32_t preset = 1, seq_col = 1, beg_col = 2, end_col = 0, meta_char = '#', line_skip = 0; };

class TabixIndex {
public:
    typedef TabixConf Conf;
    explicit TabixIndex(const Conf &c = Conf()) : conf_(c) {}

    // One data line: `name` / `pos1` (1-based) from its first two columns, the virtual offsets before and after it.  Lines must
    // come sorted by position within a sequence, the lines of a sequence together (as the files are written).
    void add_line(const std::string &name, int64_t pos1, uint64_t off_beg, uint64_t off_end) {
        if (names_.empty() || names_.back() != name) {
            for (const auto &n : names_)
                if (n == name) throw std::runtime_error("[ERROR] tabix index: the lines of sequence " + name + " are not contiguous");
            close_run();
            names_.push_back(name);
            refs_.emplace_back();
            last_pos_ = -1;
        }
        if (pos1 < 1 || pos1 > (int64_t(1) << 29)) throw std::runtime_error("[ERROR] tabix index: position outside 1 .. 2^29");
        if (pos1 < last_pos_) throw std::runtime_error("[ERROR] tabix index: lines of " + name + " are not sorted by position");
        last_pos_ = pos1;
        Ref &r = refs_.back();
        const int64_t beg = pos1 - 1, end = pos1;  // 0-based half-open: one base (end column 0: the reference's configuration)
        const uint32_t bin = reg2bin(beg, end);
        if (!run_open_ || bin != run_bin_) {
            close_run();
            run_open_ = true; run_bin_ = bin; run_beg_ = off_beg;
        }
        run_end_ = off_end;
        const size_t w = (size_t)(beg >> 14);
        if (r.linear.size() <= w) r.linear.resize(w + 1, kNoOffset);
        if (r.linear[w] == kNoOffset) r.linear[w] = off_beg;
        if (r.n_lines == 0) r.off_beg = off_beg;
        r.off_end = off_end;
        r.n_lines += 1;
    }

    // the index of a file whose data lines were all added, as a BGZF file at `path`
    void write(const std::string &path) {
        close_run();
        std::string b;
        b.append("TBI\1", 4);
        put32(b, (int32_t)names_.size());
        put32(b, conf_.preset); put32(b, conf_.seq_col); put32(b, conf_.beg_col); put32(b, conf_.end_col);
        put32(b, conf_.meta_char); put32(b, conf_.line_skip);
        size_t l_nm = 0;
        for (const auto &n : names_) l_nm += n.size() + 1;
        put32(b, (int32_t)l_nm);
        for (const auto &n : names_) b.append(n.c_str(), n.size() + 1);
        for (Ref &r : refs_) {
            put32(b, (int32_t)r.bins.size() + (r.n_lines ? 1 : 0));
            for (const auto &kv : r.bins) {
                put32(b, (int32_t)kv.first);
                put32(b, (int32_t)kv.second.size());
                for (const auto &c : kv.second) { put64(b, c.first); put64(b, c.second); }
            }
            if (r.n_lines) {  // the pseudo-bin: where the sequence's lines lie, how many there are (mapped / unmapped)
                put32(b, 37450); put32(b, 2);
                put64(b, r.off_beg); put64(b, r.off_end); put64(b, r.n_lines); put64(b, 0);
            }
            // windows without a line take the offset of the next window that has one (a query then starts no later than needed)
            uint64_t next = r.off_end;
            for (size_t w = r.linear.size(); w-- > 0;) {
                if (r.linear[w] == kNoOffset) r.linear[w] = next; else next = r.linear[w];
            }
            put32(b, (int32_t)r.linear.size());
            for (uint64_t o : r.linear) put64(b, o);
        }
        BgzfWriter w;
        w.open(path);
        w.write(b);
        w.close();
    }

    // the smallest bin that contains [beg, end): 16 kb leaves (bins 4681 ..), five levels up to the whole 2^29
    static uint32_t reg2bin(int64_t beg, int64_t end) {
        --end;
        if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
        return 0;
    }

private:
    static constexpr uint64_t kNoOffset = ~0ull;
    struct Ref {
        std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> linear;
        uint64_t off_beg = 0, off_end = 0, n_lines = 0;
    };
    void close_run() {
        if (run_open_ && !refs_.empty()) refs_.back().bins[run_bin_].emplace_back(run_beg_, run_end_);
        run_open_ = false;
    }
    static void put32(std::string &b, int32_t v) { for (int i = 0; i < 4; ++i) b.push_back((char)(((uint32_t)v >> (8 * i)) & 0xff)); }
    static void put64(std::string &b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((char)((v >> (8 * i)) & 0xff)); }
    Conf conf_;
    std::vector<std::string> names_;
    std::vector<Ref> refs_;
    bool run_open_ = false;
    uint32_t run_bin_ = 0;
    uint64_t run_beg_ = 0, run_end_ = 0;
    int64_t last_pos_ = -1;
};

// A text output that is plain (`x.vcf`) or BGZF-compressed and tabix-indexed (`x.vcf.gz` + `x.vcf.gz.tbi`), by its suffix -- as the
// reference chooses (caller.cpp:242-254).  Header lines ('#') are not indexed.
class TextOut {
public:
    void open(const std::string &path) {
        path_ = path;
        gz_ = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
        if (gz_) bg_.open(path);
        else {
            f_ = std::fopen(path.c_str(), "w");
            if (!f_) throw std::runtime_error("[ERROR] " + path + " open failure.");
        }
    }
    void write_header(const std::string &s) { put(s.data(), s.size()); }
    // whole lines, each ending in '\n', first two columns = sequence name, position
    void write_lines(const std::string &s) {
        if (!gz_) { put(s.data(), s.size()); return; }
        size_t p = 0;
        while (p < s.size()) {
            size_t e = s.find('\n', p);
            e = (e == std::string::npos) ? s.size() : e + 1;
            const uint64_t o0 = bg_.tell();
            bg_.write(s.data() + p, e - p);
            if (s[p] != '#') {
                const size_t t1 = s.find('\t', p), t2 = t1 == std::string::npos ? t1 : s.find('\t', t1 + 1);
                if (t2 == std::string::npos || t2 >= e) throw std::runtime_error("[ERROR] a data line without two columns cannot be indexed");
                idx_.add_line(s.substr(p, t1 - p), std::stoll(s.substr(t1 + 1, t2 - t1 - 1)), o0, bg_.tell());
            }
            p = e;
        }
    }
    void close() {
        if (gz_) {
            if (bg_.is_open()) { bg_.close(); idx_.write(path_ + ".tbi"); }
        } else if (f_) {
            const bool ok = std::fclose(f_) == 0;
            f_ = nullptr;
            if (!ok) throw std::runtime_error("[ERROR] write failure on " + path_);
        }
    }

private:
    void put(const char *p, size_t n) {
        if (gz_) bg_.write(p, n);
        else if (std::fwrite(p, 1, n, f_) != n) throw std::runtime_error("[ERROR] write failure on " + path_);
    }
    std::string path_;
    bool gz_ = false;
    FILE *f_ = nullptr;
    BgzfWriter bg_;
    TabixIndex idx_;
};

}  // namespace bvamd
