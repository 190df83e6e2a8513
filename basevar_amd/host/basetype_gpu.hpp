// basetype_gpu.hpp -- header-only C++17 host wrapper over the C ABI (include/basevar_amd.h).
//
// This is what a maintainer of the reference would include in src/basetype_caller.cpp: it
// mirrors the per-site interface of the reference, batched.
//
//   reference (per site, src/basetype_caller.cpp:742-743, 1113-1164)      here (per batch of sites)
//   ---------------------------------------------------------------      ---------------------------------
//   BatchInfo bi; ... fill from batchfile lines                            SlabBuilder::add_site(bi)
//   BaseType bt(&bi, min_af); bt.lrt();                                    BaseTypeEngine::lrt(builder) -> batch
//   bt.get_alt_bases() / get_lrt_af(b) / get_var_qual()                    batch.get_alt_bases(i) / get_lrt_af(i,b) / ...
//   strand_bias(...), ref_vs_alt_ranksumtest(...)                          batch.strand_bias(i, flavour), batch.rank_sums(i)
//
// Errors keep the reference's type (std::runtime_error) and, where the reference has one,
// its message (src/basetype.cpp:54-56, 113-115, 272; src/basetype.h:133-149).
#pragma once

#include <cctype>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/basevar_amd.h"

namespace bvamd {

static const std::vector<char> BASES = {'A', 'C', 'G', 'T'};  // src/basetype.h:19

inline int base_code(char b) {
    switch (b) {
        case 'A': return BV_BASE_A;
        case 'C': return BV_BASE_C;
        case 'G': return BV_BASE_G;
        case 'T': return BV_BASE_T;
        default: return BV_BASE_OTHER;
    }
}

// Packs reference-style per-site inputs (any type with the members of `struct BatchInfo`,
// src/basetype.h:25-43) into the SoA planes of bv_slab.  Host memory; the engine stages it.
class SlabBuilder {
public:
    explicit SlabBuilder(uint32_t n_samples) : n_(n_samples), pitch_((n_samples + 255u) / 256u * 256u) {}

    template <class BatchInfoLike>
    void add_site(const BatchInfoLike &bi) {
        if (bi.align_bases.size() != n_ || bi.align_base_quals.size() != n_ || bi.mapqs.size() != n_ ||
            bi.map_strands.size() != n_ || bi.base_pos_ranks.size() != n_)
            throw std::runtime_error("[ERROR] Something is wrong in batchfiles.");  // caller.cpp:736
        plain_only();
        // The reference's order of complaints (basetype_caller.cpp:738-743): _out_cvg_line -> strand_bias runs BEFORE the
        // BaseType constructor, over every sample whose token does not start with N / + / - -- an empty token (its [0] is the
        // terminator) and characters outside ACGT included -- and refuses a strand that is neither + nor - (basetype.cpp:253-273);
        // only a position that passes gets the constructor's checks.  (Held against the reference's own _basevar_caller by
        // tests/test_host_formats.py.)
        for (uint32_t i = 0; i < n_; ++i) {
            const std::string &tok = bi.align_bases[i];
            const char fb = tok.empty() ? '\0' : tok[0];
            if (fb == 'N' || fb == '+' || fb == '-') continue;
            const char s = bi.map_strands[i];
            if (s != '+' && s != '-') throw std::runtime_error(std::string("[ERROR] Get strange strand symbol: ") + s);
        }
        const size_t off = bs_.size();
        bs_.resize(off + pitch_, BV_CELL_N);
        q_.resize(off + pitch_, 0);
        mq_.resize(off + pitch_, 0);
        rp_.resize(off + pitch_, 0);
        struct Undo {  // a refused site leaves nothing behind
            SlabBuilder &sb; size_t off; bool keep = false;
            ~Undo() { if (!keep) { sb.bs_.resize(off); sb.q_.resize(off); sb.mq_.resize(off); sb.rp_.resize(off); } }
        } undo{*this, off};
        for (uint32_t i = 0; i < n_; ++i) {
            const std::string &tok = bi.align_bases[i];
            const char fb = tok.empty() ? '\0' : tok[0];  // (an empty token fails the size() != 1 check below, as in the reference)
            uint8_t cell;
            if (fb == 'N') {
                cell = BV_CELL_N;
            } else if (fb == '+') {
                cell = BV_CELL_INS;
            } else if (fb == '-') {
                cell = BV_CELL_DEL;
            } else {
                if (tok.size() != 1)  // src/basetype.cpp:54-56
                    throw std::runtime_error("[ERROR] Why dose the size of aligned base is not 1? Check: " + tok);
                const int c = base_code(fb);
                if (c == BV_BASE_OTHER) {
                    // A single character outside ACGTN+- : the reference would count it in total_depth and give it
                    // an all-eps/3 likelihood row (src/basetype.cpp:58-64).  Its own pileup never writes one
                    // (bam_record.h:28-31, caller.cpp:1060-1077) and the slab has no code for it, so such a token is
                    // refused loudly instead of being dropped silently.
                    throw std::runtime_error(std::string("[ERROR] base character '") + fb +
                                             "' is outside ACGTN+-: not representable in the slab (the reference would "
                                             "count it in the depth)");
                } else {
                    cell = (uint8_t)(c | (bi.map_strands[i] == '-' ? BV_CELL_REV : 0));  // (+ or -: checked above)
                }
            }
            bs_[off + i] = cell;
            q_[off + i] = (uint8_t)(bi.align_base_quals[i] - 33);  // src/basetype.cpp:47
            mq_[off + i] = (uint8_t)bi.mapqs[i];
            rp_[off + i] = (uint16_t)bi.base_pos_ranks[i];
        }
        ref_.push_back((uint8_t)base_code((char)std::toupper((unsigned char)(bi.ref_base.empty() ? 'N' : bi.ref_base[0]))));
        undo.keep = true;
    }

    // One site straight from per-sample planes in the slab's own encoding (e.g. a row of a PileupTile, pileup.hpp).
    void add_row(const uint8_t *cell, const uint8_t *phred, const uint8_t *mapq, const uint16_t *rank, uint8_t ref_code) {
        plain_only();
        const size_t off = bs_.size();
        bs_.resize(off + pitch_, BV_CELL_N);
        q_.resize(off + pitch_, 0);
        mq_.resize(off + pitch_, 0);
        rp_.resize(off + pitch_, 0);
        std::memcpy(&bs_[off], cell, n_);
        std::memcpy(&q_[off], phred, n_);
        std::memcpy(&mq_[off], mapq, n_);
        std::memcpy(&rp_[off], rank, (size_t)n_ * sizeof(uint16_t));
        ref_.push_back(ref_code);
    }
    // A row filled in place (batchfile_fast.hpp): begin_row() appends an all-'N' row and hands out its planes; commit_row()
    // makes it a site, drop_row() takes it back.
    struct Row {
        uint8_t *cell, *phred, *mapq;
        uint16_t *rank;
    };
    Row begin_row() {
        plain_only();
        const size_t off = (size_t)n_sites() * pitch_;
        bs_.resize(off + pitch_, BV_CELL_N);
        q_.resize(off + pitch_, 0);
        mq_.resize(off + pitch_, 0);
        rp_.resize(off + pitch_, 0);
        std::memset(&bs_[off], BV_CELL_N, pitch_);
        std::memset(&q_[off], 0, pitch_);
        std::memset(&mq_[off], 0, pitch_);
        std::memset(&rp_[off], 0, pitch_ * sizeof(uint16_t));
        return Row{&bs_[off], &q_[off], &mq_[off], &rp_[off]};
    }
    void commit_row(uint8_t ref_code) { ref_.push_back(ref_code); }
    void drop_row() {
        const size_t off = (size_t)n_sites() * pitch_;
        bs_.resize(off); q_.resize(off); mq_.resize(off); rp_.resize(off);
    }
    // All rows of `o` (same sample count), behind this builder's: blocks of sites parsed by several threads into builders
    // of their own are joined in site order this way (bv_call).
    void append(const SlabBuilder &o) {
        if (o.n_ != n_) throw std::runtime_error("[ERROR] SlabBuilder::append: different sample counts");
        plain_only(); o.plain_only();
        bs_.insert(bs_.end(), o.bs_.begin(), o.bs_.end());
        q_.insert(q_.end(), o.q_.begin(), o.q_.end());
        mq_.insert(mq_.end(), o.mq_.begin(), o.mq_.end());
        rp_.insert(rp_.end(), o.rp_.begin(), o.rp_.end());
        ref_.insert(ref_.end(), o.ref_.begin(), o.ref_.end());
    }
    // rows [first, first + k) of `o` only
    void append_rows(const SlabBuilder &o, size_t first, size_t k) {
        if (o.n_ != n_) throw std::runtime_error("[ERROR] SlabBuilder::append: different sample counts");
        plain_only(); o.plain_only();
        const size_t a = first * pitch_, b = (first + k) * pitch_;
        bs_.insert(bs_.end(), o.bs_.begin() + a, o.bs_.begin() + b);
        q_.insert(q_.end(), o.q_.begin() + a, o.q_.begin() + b);
        mq_.insert(mq_.end(), o.mq_.begin() + a, o.mq_.begin() + b);
        rp_.insert(rp_.end(), o.rp_.begin() + a, o.rp_.begin() + b);
        ref_.insert(ref_.end(), o.ref_.begin() + first, o.ref_.begin() + first + k);
    }
    void reserve_rows(size_t rows) {
        bs_.reserve(rows * pitch_); q_.reserve(rows * pitch_); mq_.reserve(rows * pitch_); rp_.reserve(rows * pitch_); ref_.reserve(rows);
    }
    const uint8_t *cell_row(size_t site) const { return &bs_[site * pitch_]; }
    const uint8_t *phred_row(size_t site) const { return &q_[site * pitch_]; }
    const uint8_t *mapq_row(size_t site) const { return &mq_[site * pitch_]; }
    const uint16_t *rank_row(size_t site) const { return &rp_[site * pitch_]; }
    uint8_t ref_code(size_t site) const { return ref_[site]; }

    void set_groups(const std::vector<uint8_t> &group_id, uint32_t n_groups) {
        gid_ = group_id;
        gid_.resize(pitch_, BV_NO_GROUP);
        n_groups_ = n_groups;
    }
    void clear() { bs_.clear(); q_.clear(); mq_.clear(); rp_.clear(); ref_.clear(); layout_ = 0; }
    // The producer's choice of the rank plane's layout (include/basevar_amd.h, BV_SLAB_RPR_TAGGED): when every read-position
    // rank of the slab is <= 8,191 -- every short-read cohort -- each rank word also takes its cell's call, and the engine's rank
    // sums (ref_vs_alt_ranksumtest, src/basetype.cpp:201-242) read mapq + rpr only.  Call it on a COMPLETE slab, before slab():
    // rows cannot be added afterwards (clear() starts a plain slab again), rank_row() then returns the tagged words.  Returns
    // whether the slab is tagged now; a slab with a longer read stays plain and is submitted as such.  Records do not depend on it.
    bool tag_ranks() {
        if (layout_ & BV_SLAB_RPR_TAGGED) return true;
        uint16_t any = 0;
        for (uint16_t v : rp_) any |= v;
        if (any & (uint16_t)~BV_RPR_TAG_MAX_RANK) return false;
        for (size_t i = 0; i < rp_.size(); ++i) rp_[i] = BV_RPR_TAGGED((uint32_t)bs_[i], (uint32_t)rp_[i]);
        layout_ = BV_SLAB_RPR_TAGGED;
        return true;
    }
    uint32_t layout() const { return layout_; }
    uint32_t n_sites() const { return (uint32_t)ref_.size(); }
    uint32_t n_samples() const { return n_; }
    uint32_t n_groups() const { return n_groups_; }

    bv_slab slab() const {
        bv_slab s{};
        s.n_sites = n_sites(); s.n_samples = n_; s.pitch = pitch_;
        s.base_strand = bs_.data(); s.qual = q_.data(); s.mapq = mq_.data(); s.rpr = rp_.data();
        s.ref_base = ref_.data();
        s.group_id = n_groups_ ? gid_.data() : nullptr;
        s.n_groups = n_groups_;
        s.mem_kind = BV_MEM_HOST;
        s.layout = layout_;
        return s;
    }

private:
    void plain_only() const {
        if (layout_) throw std::runtime_error("[ERROR] SlabBuilder: rows cannot be added to a slab whose ranks are tagged (tag_ranks)");
    }
    uint32_t n_, n_groups_ = 0, layout_ = 0;
    uint64_t pitch_;
    std::vector<uint8_t> bs_, q_, mq_, ref_, gid_;
    std::vector<uint16_t> rp_;
};

struct StrandBiasInfo {  // src/basetype.h:57-62
    int ref_fwd, ref_rev, alt_fwd, alt_rev;
    double fs, sor;
};

// Results of one batch; getters carry the reference's names (src/basetype.h:121-151).
class BaseTypeBatch {
public:
    std::vector<bv_site_result> sites;
    std::vector<bv_group_result> groups;
    uint32_t n_groups = 0;

    std::vector<char> get_alt_bases(size_t i) const {
        std::vector<char> v;
        for (int k = 0; k < sites[i].n_alt; ++k) v.push_back(BASES[sites[i].alt[k] & 3]);
        return v;
    }
    double get_lrt_af(size_t i, char b) const {
        for (int k = 0; k < sites[i].n_alt; ++k)
            if (BASES[sites[i].alt[k] & 3] == b) return sites[i].af[k];
        throw std::runtime_error(std::string("[ERROR] out_of_range:: map::at '") + b + "' not found.");
    }
    double get_var_qual(size_t i) const { return sites[i].qual; }
    int get_total_depth(size_t i) const { return (int)sites[i].total_depth; }
    double get_base_depth(size_t i, char b) const {
        const int c = base_code(b);
        if (c == BV_BASE_OTHER) throw std::runtime_error(std::string("[ERROR] out_of_range:: map::at '") + b + "' not found.");
        return (double)sites[i].depth[c];
    }
    bool has_variant(size_t i) const { return (sites[i].status & BV_SITE_VARIANT) != 0; }
    StrandBiasInfo strand_bias(size_t i, bool vcf_flavour) const {
        const bv_site_result &r = sites[i];
        const uint32_t *sb = vcf_flavour ? r.var_sb : r.cvg_sb;
        return {(int)sb[0], (int)sb[1], (int)sb[2], (int)sb[3], vcf_flavour ? r.var_fs : r.cvg_fs,
                vcf_flavour ? r.var_sor : r.cvg_sor};
    }
    // the three INFO rank sums, truncated to int exactly as the caller does (caller.cpp:1151-1157)
    int mq_rank_sum(size_t i) const { return (int)sites[i].mq_ranksum; }
    int read_pos_rank_sum(size_t i) const { return (int)sites[i].rpr_ranksum; }
    int base_q_rank_sum(size_t i) const { return (int)sites[i].bq_ranksum; }
    const bv_group_result &group(size_t i, uint32_t g) const { return groups[i * n_groups + g]; }
};

class BaseTypeEngine {
public:
    // user_min_af is the CLI value (float, src/basetype_utils.h:94); the engine applies
    // min(100/n_samples, user_min_af) in float exactly as caller.cpp:122 does.
    BaseTypeEngine(uint32_t max_sites, uint32_t n_samples, float user_min_af = 0.01f, int device = 0) {
        bv_engine_config cfg{};
        cfg.device = device; cfg.max_sites = max_sites; cfg.max_samples = n_samples;
        cfg.flags = BV_FLAG_SPARSE_TIMING;  // nobody here reads the engine's per-pass timing: spare the launches its event records
        cfg.min_af = bv_min_af(n_samples, user_min_af);
        if (bv_engine_create(&cfg, &e_) != BV_OK) throw std::runtime_error(bv_last_error(nullptr));
    }
    ~BaseTypeEngine() { if (e_) bv_engine_destroy(e_); }
    BaseTypeEngine(const BaseTypeEngine &) = delete;
    BaseTypeEngine &operator=(const BaseTypeEngine &) = delete;

    BaseTypeBatch lrt(const SlabBuilder &b) {
        BaseTypeBatch out;
        out.sites.resize(b.n_sites());
        out.n_groups = b.n_groups();
        out.groups.resize((size_t)b.n_sites() * b.n_groups());
        bv_slab s = b.slab();
        if (bv_engine_submit(e_, &s, out.sites.data(), out.n_groups ? out.groups.data() : nullptr, nullptr) != BV_OK)
            throw std::runtime_error(bv_last_error(e_));
        if (bv_engine_wait(e_) != BV_OK) throw std::runtime_error(bv_last_error(e_));
        return out;
    }

private:
    bv_engine *e_ = nullptr;
};

}  // namespace bvamd
