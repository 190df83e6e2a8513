// pileup.hpp -- the pileup that feeds the path: BAM alignments of
// one batch of samples -> per-position, per-sample first-read-wins cells -> reference-format
// batchfile rows.  Restates, step for step,
//   __create_a_batchfile          src/basetype_caller.cpp:800-874   (500 kb sub-regions, header)
//   __fetch_base_in_region        src/basetype_caller.cpp:876-939   (200 bp padding, read filter)
//   __seek_position               src/basetype_caller.cpp:941-1024  (aligned pairs, indel anchoring)
//   __write_record_to_batchfile   src/basetype_caller.cpp:1027-1101 (row text)
//   BamRecord::get_aligned_pairs  src/bam_record.cpp:217-283
// on top of bamio.hpp instead of htslib.  Quirks kept on purpose: an indel is anchored on the base to its
// left and only recorded if no earlier pair of any read -- including the same read's own match at that
// base -- already claimed the position (:1013-1019); the region test uses the un-anchored position
// (:976-977); N / S / P / H operations contribute nothing (:1004-1007).
//
// PARITY STATUS: transcribed from the cited lines; unpinned by a run of the reference binary (htslib is
// not buildable under this round's rules).  Checked against an independent Python derivation on the
// reference's own BAM fixture and on synthetic BAMs (tests/test_pileup_cpu.py); the record counts of
// SURVEY.md section 8c (real binary, 2 x range.bam: 5 VCF records, 207 CVG rows) are asserted end to end.
#pragma once

#include <zlib.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

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

// AlignBaseInfo, src/basetype_caller.h (the value of PosMap)
struct AlignBaseInfo {
    std::string ref_id;
    uint32_t ref_pos = 0;
    std::string ref_base, read_base;
    char read_base_qual = '!';
    int rpr = 0;
    int mapq = 0;
    char map_strand = '.';
};
typedef std::map<uint32_t, AlignBaseInfo> PosMap;
typedef std::vector<PosMap> PosMapVector;
typedef std::tuple<std::string, uint32_t, uint32_t> GenomeRegionTuple;  // [chr, start, end], 1-based

// ReadAlignedPair + BamRecord::get_aligned_pairs, src/bam_record.h:33-41, src/bam_record.cpp:217-283
struct ReadAlignedPair {
    int op;
    int64_t ref_pos;
    std::string ref_base;
    uint32_t qpos;
    std::string read_base, read_qual;
};
inline std::vector<ReadAlignedPair> get_aligned_pairs(const BamAlignment &al, const std::string &fa) {
    std::vector<ReadAlignedPair> pairs;
    ReadAlignedPair p;
    int64_t rpos = al.map_ref_start_pos();
    uint32_t qpos = 0;
    std::string read_qual(al.qual.size(), '!');
    for (size_t i = 0; i < al.qual.size(); ++i) read_qual[i] = (char)(al.qual[i] + 33);
    for (uint32_t c : al.cigar) {
        const int op = c & 15;
        const int64_t len = c >> 4;
        if (op == BAM_CMATCH || op == BAM_CEQUAL || op == BAM_CDIFF) {
            for (int64_t i = rpos; i < rpos + len; ++i) {
                p.op = op; p.ref_pos = i; p.ref_base = fa.substr((size_t)i, 1); p.qpos = qpos;
                p.read_base = al.seq.substr(qpos, 1); p.read_qual = read_qual.substr(qpos, 1);
                pairs.push_back(p);
                ++qpos;
            }
            rpos += len;
        } else if (op == BAM_CINS || op == BAM_CSOFT_CLIP || op == BAM_CPAD) {
            p.op = op; p.ref_pos = rpos; p.ref_base = ""; p.qpos = qpos;
            p.read_base = al.seq.substr(qpos, (size_t)len); p.read_qual = read_qual.substr(qpos, (size_t)len);
            pairs.push_back(p);
            qpos += (uint32_t)len;
        } else if (op == BAM_CDEL || op == BAM_CREF_SKIP) {
            p.op = op; p.ref_pos = rpos; p.ref_base = fa.substr((size_t)rpos, (size_t)len); p.qpos = qpos;
            p.read_base = ""; p.read_qual = "";
            pairs.push_back(p);
            rpos += len;
        }  // BAM_CHARD_CLIP: nothing
    }
    return pairs;
}

// __seek_position, src/basetype_caller.cpp:941-1024
inline void seek_position(const std::vector<BamAlignment> &reads, const std::string &fa_seq, const GenomeRegionTuple &region,
                          PosMap &sample_posinfo_map) {
    if (!sample_posinfo_map.empty())
        throw std::runtime_error("[basetype.cpp::__seek_position] 'sample_posinfo_map' must be empty.");
    const std::string &ref_id = std::get<0>(region);
    const uint32_t reg_start = std::get<1>(region), reg_end = std::get<2>(region);
    AlignBaseInfo abi;
    abi.ref_id = ref_id;
    for (const auto &al : reads) {
        abi.map_strand = al.map_strand();
        abi.mapq = al.mapq();
        const std::vector<ReadAlignedPair> pairs = get_aligned_pairs(al, fa_seq);
        const char mean_qqual_char = (char)(int(al.mean_qqual()) + 33);
        for (const auto &ap : pairs) {
            uint32_t map_ref_pos = (uint32_t)(ap.ref_pos + 1);
            if (reg_end < map_ref_pos) break;
            if (reg_start > map_ref_pos) continue;
            if (ap.op == BAM_CMATCH || ap.op == BAM_CEQUAL || ap.op == BAM_CDIFF) {
                abi.ref_base = ap.ref_base.substr(0, 1);
                abi.read_base = ap.read_base.substr(0, 1);
                abi.read_base_qual = ap.read_qual[0];
            } else if (ap.op == BAM_CINS) {
                if (!ap.ref_base.empty()) throw std::runtime_error("[ERROR] We got reference base in insertion region.");
                --map_ref_pos;  // the base left of the insertion break point
                abi.ref_base = std::string(1, fa_seq[(size_t)ap.ref_pos - 1]);
                abi.read_base = fa_seq[(size_t)ap.ref_pos - 1] + ap.read_base;
                abi.read_base_qual = mean_qqual_char;
            } else if (ap.op == BAM_CDEL) {
                if (!ap.read_base.empty()) throw std::runtime_error("[ERROR] We got read bases in deletion region.");
                --map_ref_pos;
                abi.ref_base = fa_seq[(size_t)ap.ref_pos - 1] + ap.ref_base;
                abi.read_base = std::string(1, fa_seq[(size_t)ap.ref_pos - 1]);
                abi.read_base_qual = mean_qqual_char;
            } else {
                continue;
            }
            abi.ref_pos = map_ref_pos;
            abi.rpr = (int)ap.qpos + 1;
            if (sample_posinfo_map.find(map_ref_pos) == sample_posinfo_map.end()) sample_posinfo_map.insert({map_ref_pos, abi});
        }
    }
}

// __fetch_base_in_region, src/basetype_caller.cpp:876-939.  Returns is_empty.  The samples are independent
// (one BAM, one PosMap each), so `n_threads` > 1 deals them to worker threads; the result does not depend on it.
inline void pileup_one_sample(const std::string &path, const std::string &fa_seq, int mapq_thd, const GenomeRegionTuple &region,
                              bool use_index, PosMap &sample_posinfo_map) {
    static const uint32_t REG_EXPEND_SIZE = 200;
    const std::string &ref_id = std::get<0>(region);
    const uint32_t reg_start = std::get<1>(region), reg_end = std::get<2>(region);
    const uint32_t exp_reg_start = reg_start > REG_EXPEND_SIZE ? reg_start - REG_EXPEND_SIZE : 1;
    const uint32_t exp_reg_end = reg_end + REG_EXPEND_SIZE;
    BamFile bf(path, use_index);
    // "chr:beg-end", 1-based inclusive == [beg - 1, end) 0-based
    if (bf.fetch(bf.tid_of(ref_id), (int64_t)exp_reg_start - 1, (int64_t)exp_reg_end)) {
        std::vector<BamAlignment> sample_target_reads;
        BamAlignment al;
        while (bf.next(al) >= 0) {
            if (al.mapq() < mapq_thd || al.is_duplicate() || al.is_qc_fail()) continue;
            const int64_t map_ref_start = al.map_ref_start_pos() + 1;  // 1-based
            const int64_t map_ref_end = al.map_ref_end_pos();          // 1-based
            if ((int64_t)reg_start > map_ref_end) continue;
            if ((int64_t)reg_end < map_ref_start) break;
            sample_target_reads.push_back(al);
        }
        if (!sample_target_reads.empty()) seek_position(sample_target_reads, fa_seq, region, sample_posinfo_map);
    }
}
inline bool fetch_base_in_region(const std::vector<std::string> &batch_align_files, const std::string &fa_seq, int mapq_thd,
                                 const GenomeRegionTuple &region, PosMapVector &out, bool use_index = true, int n_threads = 1) {
    const size_t base = out.size(), n = batch_align_files.size();
    out.resize(base + n);
    if (n_threads <= 1 || n < 2) {
        for (size_t i = 0; i < n; ++i) pileup_one_sample(batch_align_files[i], fa_seq, mapq_thd, region, use_index, out[base + i]);
    } else {
        std::atomic<size_t> next(0);
        std::mutex mu;
        std::string err;
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < n;) {
                try {
                    pileup_one_sample(batch_align_files[i], fa_seq, mapq_thd, region, use_index, out[base + i]);
                } catch (const std::exception &ex) {
                    std::lock_guard<std::mutex> g(mu);
                    if (err.empty()) err = ex.what();
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(work);
        for (auto &t : pool) t.join();
        if (!err.empty()) throw std::runtime_error(err);
    }
    bool is_empty = true;
    for (size_t i = 0; i < n; ++i)
        if (!out[base + i].empty()) is_empty = false;
    return is_empty;
}

// __write_record_to_batchfile, src/basetype_caller.cpp:1027-1101: appends the rows of [start, end] to `out`
inline void write_records(const PosMapVector &v, const std::string &fa_seq, const GenomeRegionTuple &region, std::string &out) {
    const std::string &ref_id = std::get<0>(region);
    const uint32_t reg_start = std::get<1>(region), reg_end = std::get<2>(region);
    const size_t sn = v.size();
    // the five per-sample columns are built as text directly (same characters as ngslib::join over the vectors
    // of :1040-1045, without an ostringstream per item)
    std::string mapq, bases, quals, ranks, strands;
    for (uint32_t pos = reg_start; pos < reg_end + 1; ++pos) {
        uint32_t depth = 0;
        mapq.clear(); ranks.clear(); bases.clear(); quals.clear(); strands.clear();
        for (size_t i = 0; i < sn; ++i) {
            if (i) { mapq += ' '; bases += ' '; quals += ' '; ranks += ' '; strands += ' '; }
            auto it = v[i].find(pos);
            if (it != v[i].end()) {
                ++depth;
                const AlignBaseInfo &a = it->second;
                if (a.ref_id != ref_id || a.ref_pos != pos) throw std::runtime_error("[ERROR] reference id or position not match.");
                mapq += std::to_string(a.mapq);
                if (a.ref_base.size() == a.read_base.size()) bases += a.read_base;
                else if (a.ref_base.size() < a.read_base.size()) { bases += '+'; bases += a.read_base; }
                else { bases += '-'; bases += a.ref_base; }
                quals += a.read_base_qual;
                ranks += std::to_string(a.rpr);
                strands += a.map_strand;
            } else {
                mapq += '0';
                bases += 'N';
                quals += '!';
                ranks += '0';
                strands += '.';
            }
        }
        out += ref_id; out += '\t'; out += std::to_string(pos); out += '\t'; out += fa_seq[pos - 1]; out += '\t';
        out += std::to_string(depth); out += '\t'; out += mapq; out += '\t'; out += bases; out += '\t'; out += quals;
        out += '\t'; out += ranks; out += '\t'; out += strands; out += '\n';
    }
}

// The same cells as one row of write_records(), but as the BatchInfo that _basevar_caller would parse back from
// that row (src/basetype_caller.cpp:688-736) -- for hosts that go from the pileup straight to the engine without
// the batchfile text.  False when no sample covers the position (the reference skips such rows, :718).
inline bool batchinfo_at(const PosMapVector &v, const std::string &fa_seq, const std::string &ref_id, uint32_t pos, BatchInfo &bi) {
    const size_t sn = v.size();
    bi = BatchInfo();
    bi.n = sn;
    bi.ref_id = ref_id;
    bi.ref_pos = pos;
    bi.ref_base = std::string(1, fa_seq[pos - 1]);
    bi.align_bases.reserve(sn); bi.align_base_quals.reserve(sn); bi.mapqs.reserve(sn);
    bi.map_strands.reserve(sn); bi.base_pos_ranks.reserve(sn);
    for (size_t i = 0; i < sn; ++i) {
        auto it = v[i].find(pos);
        if (it != v[i].end()) {
            const AlignBaseInfo &a = it->second;
            ++bi.depth;
            bi.mapqs.push_back(a.mapq);
            if (a.ref_base.size() == a.read_base.size()) bi.align_bases.push_back(a.read_base);
            else if (a.ref_base.size() < a.read_base.size()) bi.align_bases.push_back("+" + a.read_base);
            else bi.align_bases.push_back("-" + a.ref_base);
            bi.align_base_quals.push_back(a.read_base_qual);
            bi.base_pos_ranks.push_back(a.rpr);
            bi.map_strands.push_back(a.map_strand);
        } else {
            bi.mapqs.push_back(0);
            bi.align_bases.push_back("N");
            bi.align_base_quals.push_back('!');
            bi.base_pos_ranks.push_back(0);
            bi.map_strands.push_back('.');
        }
    }
    return bi.depth > 0;
}

// __create_a_batchfile, src/basetype_caller.cpp:800-874: header + rows of the whole region, walked in
// 500 kb sub-regions; `sink(text)` receives the text piecewise.  Returns has_data.
template <typename Sink>
inline bool create_a_batchfile(const std::vector<std::string> &batch_align_files, const std::vector<std::string> &batch_sample_ids,
                               const std::string &fa_seq, const GenomeRegionTuple &region, int mapq_thd, Sink sink,
                               bool use_index = true, int n_threads = 1) {
    static const uint32_t STEP_REGION_LEN = 500000;
    const std::string &ref_id = std::get<0>(region);
    const uint32_t reg_beg = std::get<1>(region), reg_end = std::get<2>(region);
    sink(batchfile_header(batch_sample_ids));
    bool has_data = false;
    for (uint32_t i = reg_beg; i < reg_end + 1; i += STEP_REGION_LEN) {
        const uint32_t sub_beg = i;
        const uint32_t sub_end = sub_beg + STEP_REGION_LEN - 1 > reg_end ? reg_end : sub_beg + STEP_REGION_LEN - 1;
        PosMapVector v;
        v.reserve(batch_align_files.size());
        const GenomeRegionTuple sub = std::make_tuple(ref_id, sub_beg, sub_end);
        const auto t0 = std::chrono::steady_clock::now();
        const bool is_empty = fetch_base_in_region(batch_align_files, fa_seq, mapq_thd, sub, v, use_index, n_threads);
        if (!is_empty) has_data = true;
        const auto t1 = std::chrono::steady_clock::now();
        std::string rows;
        write_records(v, fa_seq, sub, rows);
        sink(rows);
        if (std::getenv("BV_PILEUP_TIMING"))
            std::fprintf(stderr, "[timing] %u-%u: pileup %.2f s, rows %.2f s\n", sub_beg, sub_end,
                         std::chrono::duration<double>(t1 - t0).count(),
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    }
    return has_data;
}

}  // namespace bvamd
