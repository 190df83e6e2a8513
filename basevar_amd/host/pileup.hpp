// pileup.hpp -- the pileup that feeds the path: BAM alignments of many samples ->
// first-read-wins cells, written STRAIGHT into dense position x sample planes in the engine's own cell encoding
// (include/basevar_amd.h) -- the slab rows the engine takes, and what a batchfile row is printed from.
//
// Design (not the reference's): the reference keeps, per sample and 500 kb step, a std::map<position, AlignBaseInfo>
// of strings that it fills from a vector of per-base "aligned pair" records (src/basetype_caller.cpp:876-1024,
// src/bam_record.cpp:217-283) and later looks up once per position and sample.  Here one pass over a read's CIGAR
// claims cells directly in a PileupTile: four planes [position][sample] (call + strand, phred, mapq, read-position
// rank) plus the few indel token texts, for a window of positions sized by a cell budget (so 10^5 samples work as well
// as 10^2); samples are dealt to a thread pool, each worker writing only its samples' columns.  What a site needs
// afterwards -- the slab row, the batchfile row, the VCF/CVG text -- is read from those planes.
//
// Behaviour kept identical to the reference's, quirks included (each cited where it is implemented):
//   * reads are taken in file order and the FIRST claim of a (sample, position) wins        caller.cpp:1013-1019
//   * mapq / duplicate / QC-fail filter, reads must overlap the step                         caller.cpp:900-912
//   * the walk is bounded by the reference's own 500 kb step grid, tested on the UN-anchored position of an indel
//     (an indel whose anchor base is the last base of a step is lost; one whose anchor lies before the step claims
//     nothing that is ever printed)                                                          caller.cpp:826-846, 976-977
//   * an indel is anchored on the base to its left and recorded only if nothing -- including the same read's own
//     match at that base -- has claimed that position; its quality is int(mean read quality)  caller.cpp:982-1003
//   * N / S / P / H operations claim nothing; S and P advance the query position             bam_record.cpp:258-270
//
// PARITY STATUS: unpinned by a run of the reference binary (htslib is not buildable here).  Checked against an
// independent Python derivation on the reference's own BAM fixture and on synthetic BAMs (tests/test_pileup_cpu.py);
// the record counts of SURVEY.md section 8c (real binary, 2 x range.bam: 5 VCF records, 207 CVG rows) are asserted
// end to end.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <iterator>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <sys/resource.h>

#include <memory>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/basevar_amd.h"
#include "bamio.hpp"
#include "batchfile.hpp"

namespace bvamd {

// One chromosome of a FASTA file (plain, gzip or bgzip -- zlib reads all three); the reference gets
// the same string from faidx (src/fasta.cpp), letter case as in the file.
inline std::string load_fasta_sequence(const std::string &path, const std::string &ref_id) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("[ERROR] cannot open " + path);
    gzbuffer(f, 1 << 20);
    std::string seq, line;
    bool in_target = false, found = false;
    char tmp[1 << 16];
    while (gzgets(f, tmp, sizeof tmp)) {
        line = tmp;
        const bool complete = !line.empty() && line.back() == '\n';
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (in_target) break;
            size_t e = 1;
            while (e < line.size() && line[e] != ' ' && line[e] != '\t') ++e;
            in_target = line.substr(1, e - 1) == ref_id;
            found = found || in_target;
            // a header line longer than the buffer: drop its remainder
            bool c = complete;
            while (!c && gzgets(f, tmp, sizeof tmp)) c = std::strchr(tmp, '\n') != nullptr;
        } else if (in_target) {
            seq += line;
        }
    }
    gzclose(f);
    if (!found) throw std::runtime_error("[ERROR] " + ref_id + " not found in " + path);
    return seq;
}

typedef std::tuple<std::string, uint32_t, uint32_t> GenomeRegionTuple;  // [chr, start, end], 1-based

static const uint32_t PILEUP_STEP = 500000;   // the reference's sub-region length, caller.cpp:826
static const uint32_t PILEUP_PAD = 200;       // REG_EXPEND_SIZE, caller.cpp:883

// Dense first-read-wins cells of a window of positions [beg, end] (1-based, inclusive) for n samples.
// Row = position, column = sample; a cell is claimed <=> its rank is non-zero (ranks start at 1).
struct PileupTile {
    std::string ref_id;
    uint32_t beg = 0, end = 0;
    size_t n_samples = 0, pitch = 0;
    std::vector<uint8_t> cell;    // BV_CELL_*: base | strand for A/C/G/T; N / + / - tokens carry the strand bit too
    std::vector<uint8_t> qual;    // phred (quality character - 33)
    std::vector<uint8_t> mapq;
    std::vector<uint16_t> rank;   // read-position rank (qpos + 1)
    std::vector<uint32_t> depth;  // claimed samples per position == the batchfile's Depth column
    struct IndelToken {
        uint32_t pos, sample;
        std::string text;         // "+" + anchor base + inserted bases, or "-" + anchor base + deleted bases
    };
    std::vector<IndelToken> indels;  // sorted by (pos, sample) after pileup_tile()

    size_t rows() const { return (size_t)(end - beg) + 1; }
    size_t at(uint32_t pos, size_t sample) const { return (size_t)(pos - beg) * pitch + sample; }
    void reset(const std::string &id, uint32_t b, uint32_t e, size_t n) {
        ref_id = id; beg = b; end = e; n_samples = n; pitch = (n + 255) / 256 * 256;
        const size_t cells = rows() * pitch;
        cell.assign(cells, BV_CELL_N);
        qual.assign(cells, 0);
        mapq.assign(cells, 0);
        rank.assign(cells, 0);
        depth.assign(rows(), 0);
        indels.clear();
    }
};

inline uint8_t pileup_base_code(char c) {  // read base letter -> cell code (bamio: A C G T N, ' ' for other nibbles)
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        case 'N': return BV_CELL_N;
        default:  // the reference prints such a letter into the batchfile and then fails to parse it back (basetype.cpp:54-56)
            throw std::runtime_error(std::string("[ERROR] Why dose the size of aligned base is not 1? Check: ") + c);
    }
}

// One read of one sample: walk its CIGAR once and claim cells.  [gb, ge]: the reference's step (500 kb grid) that
// bounds the walk; cells outside the tile's own window are not stored.
inline void pileup_claim_read(const BamAlignment &al, const std::string &fa, uint32_t gb, uint32_t ge, size_t sample,
                              PileupTile &t, std::vector<uint8_t> &seen, std::vector<PileupTile::IndelToken> &indels) {
    const uint8_t strand = (al.flag & BAM_FREVERSE) ? BV_CELL_REV : 0;
    const uint8_t mq = (uint8_t)al.mapq();
    const uint8_t mean_q = (uint8_t)(int)al.mean_qqual();  // char(int(mean) + 33) in the reference, caller.cpp:963
    int64_t rpos = al.pos;   // 0-based reference position of the next reference-consuming base
    uint32_t qpos = 0;
    auto claim = [&](int64_t pos1, uint8_t code, uint8_t q, uint32_t rk) -> bool {
        if (pos1 < (int64_t)t.beg || pos1 > (int64_t)t.end) return false;
        uint8_t &s = seen[(size_t)(pos1 - t.beg)];
        if (s) return false;  // first read wins
        s = 1;
        const size_t k = t.at((uint32_t)pos1, sample);
        t.cell[k] = code; t.qual[k] = q; t.mapq[k] = mq; t.rank[k] = (uint16_t)rk;
        __atomic_fetch_add(&t.depth[(size_t)(pos1 - t.beg)], 1u, __ATOMIC_RELAXED);
        return true;
    };
    for (uint32_t c : al.cigar) {
        const int op = c & 15;
        const int64_t len = c >> 4;
        if (op == BAM_CMATCH || op == BAM_CEQUAL || op == BAM_CDIFF) {
            for (int64_t i = 0; i < len; ++i) {
                const int64_t pos1 = rpos + i + 1;
                if ((int64_t)ge < pos1) return;      // beyond the step: the rest of the read too (caller.cpp:976)
                if ((int64_t)gb > pos1) continue;    // before the step (caller.cpp:977)
                const uint8_t b = pileup_base_code(al.seq[qpos + (uint32_t)i]);
                claim(pos1, (uint8_t)(b | strand), al.qual[qpos + (uint32_t)i], qpos + (uint32_t)i + 1);
            }
            rpos += len; qpos += (uint32_t)len;
        } else if (op == BAM_CINS || op == BAM_CDEL) {
            const int64_t pos1 = rpos + 1;           // the UN-anchored position is what the step test sees
            if ((int64_t)ge < pos1) return;
            if ((int64_t)gb <= pos1 && rpos >= 1) {
                // anchored on the base to the left of the break point (caller.cpp:982-1003)
                if (claim(pos1 - 1, (uint8_t)((op == BAM_CINS ? BV_CELL_INS : BV_CELL_DEL) | strand), mean_q, qpos + 1)) {
                    std::string text(1, op == BAM_CINS ? '+' : '-');
                    text += fa[(size_t)rpos - 1];
                    if (op == BAM_CINS) text += al.seq.substr(qpos, (size_t)len);
                    else text += fa.substr((size_t)rpos, (size_t)len);
                    indels.push_back({(uint32_t)(pos1 - 1), (uint32_t)sample, std::move(text)});
                }
            }
            if (op == BAM_CINS) qpos += (uint32_t)len; else rpos += len;
        } else if (op == BAM_CREF_SKIP) {
            if ((int64_t)ge < rpos + 1) return;
            rpos += len;
        } else if (op == BAM_CSOFT_CLIP || op == BAM_CPAD) {
            if ((int64_t)ge < rpos + 1) return;
            qpos += (uint32_t)len;  // the reference advances the query on P as on S (bam_record.cpp:258-265)
        }  // BAM_CHARD_CLIP: nothing
    }
}

// The samples' BAM readers, kept open across windows: opening one parses the header and loads the .bai, which the
// reference pays per 500 kb step and file (caller.cpp:876-900) and round 2 paid per WINDOW and file -- at 10^5 samples a
// window is 64 positions.  As many readers stay open as the process may hold descriptors (RLIMIT_NOFILE less a margin);
// the samples beyond that are opened per window as before.  A sample is touched by one worker at a time, so its
// reader needs no lock.
class BamPool {
public:
    BamPool(const std::vector<std::string> &paths, bool use_index) : paths_(paths), use_index_(use_index), open_(paths.size()) {
        struct rlimit rl;
        size_t lim = 256;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur != RLIM_INFINITY) lim = (size_t)rl.rlim_cur;
        else if (getrlimit(RLIMIT_NOFILE, &rl) == 0) lim = 1u << 20;
        keep_ = lim > 96 ? std::min(paths.size(), lim - 96) : 0;
        // ... and at most 4,096 by default: a kept reader holds its loaded index and BGZF buffers (up to a few MB each), and
        // where RLIMIT_NOFILE is 2^20 or unlimited a cohort of 10^5 samples would otherwise keep tens of GB of them.
        // BASEVAR_AMD_BAM_KEEP=<n> sets the number explicitly (0 = reopen per window; measurements, or hosts with memory to spare).
        keep_ = std::min(keep_, (size_t)4096);
        if (const char *k = std::getenv("BASEVAR_AMD_BAM_KEEP")) keep_ = std::min(paths.size(), (size_t)std::strtoul(k, nullptr, 10));
    }
    struct Handle {
        BamFile *bf;
        std::unique_ptr<BamFile> own;  // a reader of this window only (beyond the descriptors the pool may keep)
    };
    Handle get(size_t i) {
        if (i < keep_) {
            if (!open_[i]) open_[i].reset(new BamFile(paths_[i], use_index_));
            return Handle{open_[i].get(), nullptr};
        }
        Handle h{nullptr, std::unique_ptr<BamFile>(new BamFile(paths_[i], use_index_))};
        h.bf = h.own.get();
        return h;
    }
    size_t size() const { return paths_.size(); }
    size_t kept() const { return keep_; }
private:
    const std::vector<std::string> &paths_;
    bool use_index_;
    size_t keep_ = 0;
    std::vector<std::unique_ptr<BamFile>> open_;
};

// All reads of one sample that touch the tile.  [gb, ge] as above.
inline void pileup_one_sample(BamPool &pool, const std::string &fa, int mapq_thd, uint32_t gb, uint32_t ge,
                              size_t sample, PileupTile &t, std::vector<uint8_t> &seen, std::vector<PileupTile::IndelToken> &indels) {
    std::fill(seen.begin(), seen.end(), (uint8_t)0);
    // the reference fetches its step +- 200 bp and keeps the reads that overlap the step; for a window inside the
    // step the reads that can claim one of its cells are those that overlap the window (+ 1 for a left anchor)
    const uint32_t lo = t.beg > PILEUP_PAD ? t.beg - PILEUP_PAD : 1, hi = t.end + PILEUP_PAD;
    BamPool::Handle h = pool.get(sample);
    BamFile &bf = *h.bf;
    if (!bf.fetch(bf.tid_of(t.ref_id), (int64_t)lo - 1, (int64_t)hi)) return;
    BamAlignment al;
    while (bf.next(al) >= 0) {
        if (al.mapq() < mapq_thd || al.is_duplicate() || al.is_qc_fail()) continue;  // caller.cpp:906
        const int64_t first = al.map_ref_start_pos() + 1, last = al.map_ref_end_pos();  // 1-based, inclusive
        if ((int64_t)gb > last) continue;    // caller.cpp:909-910, against the step
        if ((int64_t)ge < first) break;
        if ((int64_t)t.end + 1 < first) break;        // nothing of this or any later read can reach the window
        if ((int64_t)t.beg > last + 1) continue;      // (an insertion right after the read's last base anchors at `last`)
        pileup_claim_read(al, fa, gb, ge, sample, t, seen, indels);
    }
}

// The pileup of every sample over [beg, end] into `t`.  `region_beg`: where the caller's whole region starts -- the
// reference's steps are laid out from there in units of 500 kb, and [beg, end] must lie inside one of them.
inline void pileup_tile(BamPool &bams, const std::string &fa, const std::string &ref_id, uint32_t region_beg,
                        uint32_t region_end, uint32_t beg, uint32_t end, int mapq_thd, int n_threads, PileupTile &t) {
    const uint32_t gb = region_beg + (beg - region_beg) / PILEUP_STEP * PILEUP_STEP;
    const uint32_t ge = std::min(region_end, gb + PILEUP_STEP - 1);
    if (end > ge) throw std::runtime_error("[pileup_tile] a window must not cross the 500 kb step grid");
    t.reset(ref_id, beg, end, bams.size());
    const size_t n = bams.size();
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), n));
    std::vector<std::vector<PileupTile::IndelToken>> found((size_t)nt);
    std::atomic<size_t> next(0);
    std::mutex mu;
    std::string err;
    auto work = [&](int w) {
        std::vector<uint8_t> seen(t.rows());
        for (size_t i; (i = next.fetch_add(1)) < n;) {
            try {
                pileup_one_sample(bams, fa, mapq_thd, gb, ge, i, t, seen, found[(size_t)w]);
            } catch (const std::exception &ex) {
                std::lock_guard<std::mutex> g(mu);
                if (err.empty()) err = ex.what();
            }
        }
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int w = 0; w < nt; ++w) pool.emplace_back(work, w);
        for (auto &th : pool) th.join();
    }
    if (!err.empty()) throw std::runtime_error(err);
    for (auto &f : found) t.indels.insert(t.indels.end(), std::make_move_iterator(f.begin()), std::make_move_iterator(f.end()));
    std::sort(t.indels.begin(), t.indels.end(), [](const PileupTile::IndelToken &a, const PileupTile::IndelToken &b) {
        return a.pos != b.pos ? a.pos < b.pos : a.sample < b.sample;
    });
}

// Window length for n samples: the reference's whole step when it fits a cell budget, else what the budget allows.
inline uint32_t pileup_window(size_t n_samples, size_t cell_budget = (size_t)1 << 27) {
    const size_t pitch = (n_samples + 255) / 256 * 256;
    return (uint32_t)std::max<size_t>(64, std::min<size_t>(PILEUP_STEP, cell_budget / std::max<size_t>(pitch, 1)));
}

// Calls fn(tile) for consecutive windows that cover [beg, end]; windows never cross the reference's step grid.
template <typename Fn>
inline void pileup_region(const std::vector<std::string> &bams, const std::string &fa, const std::string &ref_id, uint32_t beg,
                          uint32_t end, int mapq_thd, bool use_index, int n_threads, Fn fn, uint32_t window = 0) {
    if (window == 0) window = pileup_window(bams.size());
    BamPool pool(bams, use_index);
    PileupTile t;
    for (uint32_t sb = beg; sb <= end; sb += PILEUP_STEP) {
        const uint32_t se = std::min(end, sb + PILEUP_STEP - 1);
        for (uint32_t wb = sb; wb <= se; wb += window) {
            const uint32_t we = std::min(se, wb + window - 1);
            pileup_tile(pool, fa, ref_id, beg, end, wb, we, mapq_thd, n_threads, t);
            fn(t);
            if (we == UINT32_MAX) return;
        }
        if (se == UINT32_MAX) return;
    }
}

// The text of the tile's rows in the reference's batchfile format, every position of the window, covered or not
// (__write_record_to_batchfile, caller.cpp:1027-1101).
inline void pileup_rows_text(const PileupTile &t, const std::string &fa, std::string &out) {
    static const char LETTER[4] = {'A', 'C', 'G', 'T'};
    std::string mapq, bases, quals, ranks, strands;
    size_t next_indel = 0;
    for (uint32_t pos = t.beg; pos <= t.end; ++pos) {
        mapq.clear(); bases.clear(); quals.clear(); ranks.clear(); strands.clear();
        for (size_t i = 0; i < t.n_samples; ++i) {
            if (i) { mapq += ' '; bases += ' '; quals += ' '; ranks += ' '; strands += ' '; }
            const size_t k = t.at(pos, i);
            if (t.rank[k] == 0) {  // unclaimed: the reference's placeholders
                mapq += '0'; bases += 'N'; quals += '!'; ranks += '0'; strands += '.';
                continue;
            }
            const uint8_t c = t.cell[k];
            mapq += std::to_string((int)t.mapq[k]);
            if (!(c & BV_CELL_NOCALL)) bases += LETTER[c & 3];
            else if ((c & 3) == 0) bases += 'N';
            else {
                while (next_indel < t.indels.size() && (t.indels[next_indel].pos < pos ||
                       (t.indels[next_indel].pos == pos && t.indels[next_indel].sample < i))) ++next_indel;
                bases += t.indels[next_indel].text;
            }
            quals += (char)(t.qual[k] + 33);
            ranks += std::to_string((int)t.rank[k]);
            strands += (c & BV_CELL_REV) ? '-' : '+';
        }
        out += t.ref_id; out += '\t'; out += std::to_string(pos); out += '\t'; out += fa[pos - 1]; out += '\t';
        out += std::to_string(t.depth[pos - t.beg]); out += '\t'; out += mapq; out += '\t'; out += bases; out += '\t';
        out += quals; out += '\t'; out += ranks; out += '\t'; out += strands; out += '\n';
    }
}

// Batchfile creation (__create_a_batchfile, caller.cpp:800-874): header + the rows of the whole region; `sink(text)`
// receives the text piecewise.  Returns has_data.
template <typename Sink>
inline bool create_a_batchfile(const std::vector<std::string> &batch_align_files, const std::vector<std::string> &batch_sample_ids,
                               const std::string &fa_seq, const GenomeRegionTuple &region, int mapq_thd, Sink sink,
                               bool use_index = true, int n_threads = 1, uint32_t window = 0) {
    sink(batchfile_header(batch_sample_ids));
    bool has_data = false;
    pileup_region(batch_align_files, fa_seq, std::get<0>(region), std::get<1>(region), std::get<2>(region), mapq_thd, use_index,
                  n_threads, [&](const PileupTile &t) {
                      for (uint32_t d : t.depth) has_data = has_data || d != 0;
                      std::string rows;
                      pileup_rows_text(t, fa_seq, rows);
                      sink(rows);
                  }, window);
    return has_data;
}

}  // namespace bvamd
