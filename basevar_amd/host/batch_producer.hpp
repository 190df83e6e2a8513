// batch_producer.hpp -- reference-format batchfiles -> slab rows, on T host threads.
//
// The reference joins one row from each of its NB batchfiles per position on ONE thread per region
// (src/basetype_caller.cpp:586-611); at 10,000 samples per position that thread parses ~1,000 positions a second, six orders
// of magnitude below the engine.  Here the same rows go through a pipeline of BLOCKS of R consecutive positions:
//
//   read   (task per file and block)   inflate + split R lines of one file -- a gzip stream is sequential, so a file's blocks
//                                      are read one after the other, but the NB files of a block, and blocks k + 1, k + 2 of
//                                      other files, are read at the same time.  A BGZF file (what the reference writes and
//                                      requires, src/basetype_caller.cpp:428) is not one stream but a chain of independent
//                                      <= 64 KiB gzip members: there the file is FETCHED in segments of 16 members (raw
//                                      bytes, sequential, cheap), the segments are INFLATED by tasks of their own, several of
//                                      one file at a time, and only the cutting of lines is sequential per file -- one plain
//                                      gzip stream inflates ~40,000 rows of 200 samples a second, which capped every thread
//                                      count above 16 (round 5, profiles/r5_host_pipeline.txt)
//   parse  (task per chunk of a block) the byte-level reader (batchfile_fast.hpp) on a run of consecutive positions, one row
//                                      from every file each, into a slab builder of its own
//   join   (the caller's thread)       the chunks of a block in position order -> sink(part, texts); the block's line storage
//                                      goes back to the readers
//
// with K block buffers in flight, so that reading block k + 2, parsing block k + 1 and joining block k overlap (round 3 ran
// the three as phases with a barrier between them and started its threads anew in each: 3.6 x on 16 threads).  Same rows, and
// the error of the first offending position in position order, as the position-by-position loop: what precedes the
// offending position is delivered, then the error is thrown on the caller's thread.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "batchfile_fast.hpp"

namespace bvamd {

// Lines of a gzip / BGZF / plain file (zlib reads all three): gzread into a large buffer, lines cut with memchr.
class GzLineReader {
public:
    GzLineReader() = default;
    GzLineReader(const GzLineReader &) = delete;
    GzLineReader &operator=(const GzLineReader &) = delete;
    ~GzLineReader() { if (f_) gzclose(f_); }
    bool open(const std::string &path) {
        f_ = gzopen(path.c_str(), "rb");
        if (f_) gzbuffer(f_, 1 << 20);
        buf_.resize(kBuf);
        return f_ != nullptr;
    }
    // the next line without its '\n' (a last line without one counts); false at the end of the file
    bool getline(std::string &line) {
        line.clear();
        for (;;) {
            if (pos_ < end_) {
                const char *b = buf_.data() + pos_;
                const char *nl = (const char *)std::memchr(b, '\n', end_ - pos_);
                if (nl) {
                    line.append(b, (size_t)(nl - b));
                    pos_ = (size_t)(nl - buf_.data()) + 1;
                    return true;
                }
                line.append(b, end_ - pos_);
                pos_ = end_;
            }
            if (eof_) return !line.empty();
            const int n = gzread(f_, &buf_[0], (unsigned)kBuf);
            if (n <= 0) {
                // the end of the file -- or of what zlib could make of it: a damaged or truncated gzip member must not read as a
                // shorter file (the BGZF path says "truncated BGZF member" for the same damage)
                int errnum = Z_OK;
                const char *msg = gzerror(f_, &errnum);
                if (n < 0 || (errnum != Z_OK && errnum != Z_STREAM_END))
                    throw std::runtime_error(std::string("[ERROR] damaged or truncated gzip input: ") + (msg && *msg ? msg : "read error"));
                eof_ = true; pos_ = end_ = 0; continue;
            }
            pos_ = 0; end_ = (size_t)n;
        }
    }
private:
    static constexpr size_t kBuf = (size_t)4 << 20;
    gzFile f_ = nullptr;
    std::string buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

// A fixed set of worker threads that run queued tasks (no task waits for another task: the bookkeeping below queues a task
// only when it can run to its end).
class TaskPool {
public:
    explicit TaskPool(int threads) {
        const int n = std::max(1, threads);
        for (int i = 0; i < n; ++i) workers_.emplace_back([this]() { loop(); });
    }
    ~TaskPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void submit(std::function<void()> fn) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(fn)); }
        cv_.notify_one();
    }
private:
    void loop() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    std::vector<std::thread> workers_;
    std::deque<std::function<void()>> q_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool stop_ = false;
};

struct ProducerClock {  // seconds summed over the pool's threads (read, parse) and of the caller's thread (join + sink)
    double read = 0, parse = 0, join = 0;
    double fetch = 0, inflate = 0, split = 0;  // BGZF files: the three parts of `read`
    size_t n_fetch = 0, n_inflate = 0, n_split = 0;
};

// One BGZF file read as segments of members: fetch (sequential) -> inflate (a task per segment) -> lines (sequential).
struct BgzfSegment {
    std::vector<unsigned char> raw;          // the members, back to back
    std::vector<uint32_t> off, clen, isize;  // per member: where its deflate data starts in raw, its length, the inflated size
    std::string text;
    size_t pos = 0;                          // bytes of text already cut into lines
    bool inflated = false;
};
struct BgzfFile {
    std::FILE *fp = nullptr;
    bool raw_eof = false, fetching = false;
    std::deque<std::unique_ptr<BgzfSegment>> segs;  // in file order
    std::vector<std::unique_ptr<BgzfSegment>> spare; // consumed segments, reused with their buffers (freeing and re-allocating
                                                     // a MiB per segment means an munmap -- a TLB shoot-down on every thread -- per
                                                     // millisecond and file: measured, it is what stopped 64 threads at 40 k rows/s)
    size_t skip_lines = 0;                   // header lines still to drop
    std::string carry;                       // the beginning of a line that continues in the next segment
    size_t k = 0;                            // lines of the file's current block delivered so far
    ~BgzfFile() { if (fp) std::fclose(fp); }
};
inline bool bgzf_member_header(const unsigned char *h) {  // SAM spec 4.1: gzip member with the 'BC' extra subfield first
    return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
}

class BatchfileProducer {
public:
    // `readers`: one open reader per batchfile, positioned behind the header; `first_row[f]` (have_row[f]): a data row that the
    // header scan has already taken from file f
    BatchfileProducer(std::vector<GzLineReader> &readers, std::vector<std::string> first_row, const std::vector<bool> &have_row, size_t n_sample,
                      int threads)
        : rd_(readers), first_row_(std::move(first_row)), have_row_(have_row.begin(), have_row.end()), n_sample_(n_sample), threads_(std::max(1, threads)) {}

    // Tell the producer where the files are and how many header lines each has: a file that turns out to be BGZF is then read
    // from its start through the segment pipeline (its reader in `readers` and its first_row are not used); anything else --
    // plain gzip, plain text -- keeps its sequential reader.  Without this call every file is read sequentially.
    void set_paths(const std::vector<std::string> &paths, const std::vector<size_t> &header_lines) {
        bg_.clear();
        bg_.resize(paths.size());
        for (size_t f = 0; f < paths.size() && f < rd_.size(); ++f) {
            std::FILE *fp = std::fopen(paths[f].c_str(), "rb");
            if (!fp) continue;
            unsigned char h[18];
            const bool is_bgzf = std::fread(h, 1, 18, fp) == 18 && bgzf_member_header(h);
            if (is_bgzf && std::fseek(fp, 0, SEEK_SET) == 0) {
                bg_[f].reset(new BgzfFile);
                bg_[f]->fp = fp;
                bg_[f]->skip_lines = f < header_lines.size() ? header_lines[f] : 0;
            } else {
                std::fclose(fp);
            }
        }
    }
    size_t bgzf_files() const { size_t n = 0; for (const auto &b : bg_) n += b ? 1 : 0; return n; }

    // sink(SlabBuilder &part, std::vector<SiteText> &texts): consecutive positions, in position order, on the calling thread
    // (the part is the producer's: copy what is wanted, it is reused); returns false to stop early.  Throws the first error in
    // position order after delivering what precedes it.
    template <class Sink>
    void run(Sink &&sink) {
        const size_t NB = rd_.size();
        if (NB == 0) return;
        // positions per block: ~32 MB of row text, at least a few chunks per thread
        R_ = std::max<size_t>(std::max<size_t>(64, 4 * (size_t)threads_), std::min<size_t>(4096, ((size_t)1 << 25) / std::max<size_t>(n_sample_ * 12, 1)));
        chunk_ = std::max<size_t>(1, std::min<size_t>(16, R_ / (4 * (size_t)threads_)));
        bg_.resize(NB);
        blocks_.resize(kBlocks);
        for (auto &b : blocks_) {
            b.lines.assign(NB, std::vector<std::string>(R_));
            b.got.assign(NB, 0);
        }
        file_next_.assign(NB, 0);
        file_busy_.assign(NB, false);
        TaskPool pool(threads_);
        pool_ = &pool;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t f = 0; f < NB; ++f) try_read(f);
        }
        bool go_on = true;
        std::exception_ptr err;
        for (size_t b = 0;; ++b) {
            Block &B = blocks_[b % kBlocks];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return B.index == b && B.parsed; });
            }
            const double t0 = now();
            const bool last = B.n_blk < R_;
            // the chunks in position order; the first error in position order ends the run
            for (Part &p : B.parts) {
                if (go_on && !err && p.slab && p.slab->n_sites()) go_on = sink(*p.slab, p.text);
                if (p.error && !err) err = p.error;
                p.text.clear(); p.error = nullptr;
            }
            clock.join += now() - t0;
            {
                std::lock_guard<std::mutex> lk(mu_);
                for (Part &p : B.parts)  // the builders go back with their buffers (no free / allocate of ~0.5 MB per chunk)
                    if (p.slab) { p.slab->clear(); spare_slabs_.push_back(std::move(p.slab)); }
                B.parsed = false;
                B.index = (size_t)-1;
                freed_ = b + 1;
                if (last || err || !go_on) stop_at_ = std::min(stop_at_, b);  // nothing beyond this block is wanted
                else for (size_t f = 0; f < NB; ++f) try_read(f);
            }
            if (last || err || !go_on) break;
        }
        {
            // tasks still in flight (reads of later blocks) touch this object: let them finish
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return in_flight_ == 0; });
        }
        pool_ = nullptr;
        if (err) std::rethrow_exception(err);
    }

    ProducerClock clock;
    size_t block_sites() const { return R_; }

private:
    struct Part {
        std::unique_ptr<SlabBuilder> slab;
        std::vector<SiteText> text;
        std::exception_ptr error;
    };
    struct Block {
        size_t index = (size_t)-1;                    // which block of the job this buffer holds
        std::vector<std::vector<std::string>> lines;  // [file][R]
        std::vector<size_t> got;                      // lines file f delivered
        size_t reads_done = 0, parts_done = 0, n_blk = 0;
        std::vector<Part> parts;
        bool parsed = false;
    };
    static constexpr size_t kBlocks = 3;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    // (mu_ held) queue the read of file f's next block if its buffer is free and nobody is reading the file
    void try_read(size_t f) {
        if (bg_[f]) {
            // a BGZF file: keep its window of segments full, and cut lines whenever the next segment is inflated (or the file is through)
            BgzfFile &F = *bg_[f];
            if (!F.fetching && !F.raw_eof && F.segs.size() < kWindow) {
                F.fetching = true;
                ++in_flight_;
                pool_->submit([this, f]() { fetch_task(f); });
            }
            const bool data = !F.segs.empty() && F.segs.front()->inflated, at_end = F.raw_eof && F.segs.empty() && !F.fetching;
            if (!data && !at_end) return;
        }
        const size_t b = file_next_[f];
        if (file_busy_[f] || b >= freed_ + kBlocks || b > stop_at_) return;
        Block &B = blocks_[b % kBlocks];
        if (B.index == (size_t)-1) { B.index = b; B.reads_done = 0; B.parts_done = 0; B.parsed = false; B.parts.clear(); }
        if (B.index != b) return;
        file_busy_[f] = true;
        ++in_flight_;
        if (bg_[f]) pool_->submit([this, f, b]() { split_task(f, b); });
        else pool_->submit([this, f, b]() { read_task(f, b); });
    }
    // ---- BGZF files
    static constexpr size_t kWindow = 6;      // segments of one file in flight (fetched, being inflated, waiting to be cut)
    static constexpr size_t kSegMembers = 16; // BGZF members per segment: <= 1 MiB of text, ~1 ms of inflate
    void fetch_task(size_t f) {
        const double t0 = now();
        BgzfFile &F = *bg_[f];
        std::unique_ptr<BgzfSegment> seg;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!F.spare.empty()) { seg = std::move(F.spare.back()); F.spare.pop_back(); }
        }
        if (!seg) seg.reset(new BgzfSegment);
        seg->raw.clear(); seg->off.clear(); seg->clen.clear(); seg->isize.clear(); seg->pos = 0; seg->inflated = false;
        bool eof = false;
        std::exception_ptr ex;
        try {
            for (size_t m = 0; m < kSegMembers; ++m) {
                unsigned char h[18];
                const size_t n = std::fread(h, 1, 18, F.fp);
                if (n == 0) { eof = true; break; }
                if (n != 18 || !bgzf_member_header(h)) throw std::runtime_error("[ERROR] not a BGZF member where one was expected (truncated or damaged batchfile)");
                const size_t total = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1, xlen = (size_t)h[10] | ((size_t)h[11] << 8);
                if (total < 12 + xlen + 8) throw std::runtime_error("[ERROR] damaged BGZF member");
                const size_t at = seg->raw.size();
                seg->raw.resize(at + total);
                std::memcpy(&seg->raw[at], h, 18);
                if (std::fread(&seg->raw[at + 18], 1, total - 18, F.fp) != total - 18) throw std::runtime_error("[ERROR] truncated BGZF member");
                const unsigned char *t = &seg->raw[at + total - 4];
                seg->off.push_back((uint32_t)(at + 12 + xlen));
                seg->clen.push_back((uint32_t)(total - 12 - xlen - 8));
                seg->isize.push_back((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24));
            }
        } catch (...) { ex = std::current_exception(); eof = true; }
        const double dt = now() - t0;
        std::lock_guard<std::mutex> lk(mu_);
        clock.read += dt; clock.fetch += dt; ++clock.n_fetch;
        if (ex && !read_error_) read_error_ = ex;
        if (ex) seg->off.clear();
        if (eof) F.raw_eof = true;
        F.fetching = false;
        if (!seg->off.empty()) {
            BgzfSegment *sp = seg.get();
            F.segs.push_back(std::move(seg));
            ++in_flight_;
            pool_->submit([this, f, sp]() { inflate_task(f, sp); });
        }
        try_read(f);
        --in_flight_;
        cv_.notify_all();
    }
    void inflate_task(size_t f, BgzfSegment *seg) {
        const double t0 = now();
        std::exception_ptr ex;
        try {
            size_t total = 0;
            for (uint32_t n : seg->isize) total += n;
            seg->text.resize(total);
            z_stream zs;
            std::memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("[ERROR] inflateInit2 failed");
            size_t at = 0;
            for (size_t m = 0; m < seg->off.size(); ++m) {
                if (seg->isize[m] == 0) continue;  // (the end-of-file marker, or an empty member)
                inflateReset(&zs);
                zs.next_in = &seg->raw[seg->off[m]];
                zs.avail_in = seg->clen[m];
                zs.next_out = reinterpret_cast<Bytef *>(&seg->text[at]);
                zs.avail_out = seg->isize[m];
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); throw std::runtime_error("[ERROR] a BGZF member does not inflate to its recorded size"); }
                at += seg->isize[m];
            }
            inflateEnd(&zs);
        } catch (...) { ex = std::current_exception(); seg->text.clear(); }
        const double dt = now() - t0;
        std::lock_guard<std::mutex> lk(mu_);
        clock.read += dt; clock.inflate += dt; ++clock.n_inflate;
        if (ex && !read_error_) read_error_ = ex;
        seg->inflated = true;
        try_read(f);
        --in_flight_;
        cv_.notify_all();
    }
    // the lines of file f's block b from its inflated segments, as far as they reach: the block is complete at R lines or at the
    // end of the file; else the task ends and is queued again when the next segment is inflated
    void split_task(size_t f, size_t b) {
        const double t0 = now();
        BgzfFile &F = *bg_[f];
        Block &B = blocks_[b % kBlocks];
        auto deliver = [&](const char *p, size_t n) {
            if (F.skip_lines) { --F.skip_lines; F.carry.clear(); return; }
            std::string &dst = B.lines[f][F.k++];
            if (F.carry.empty()) dst.assign(p, n);
            else { F.carry.append(p, n); dst.swap(F.carry); F.carry.clear(); }
        };
        bool at_end = false;
        for (;;) {
            BgzfSegment *s = nullptr;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!F.segs.empty() && F.segs.front()->inflated) s = F.segs.front().get();
                else at_end = F.raw_eof && F.segs.empty() && !F.fetching;
            }
            if (!s) break;
            const char *base = s->text.data();
            const size_t end = s->text.size();
            while (F.k < R_ && s->pos < end) {
                const char *nl = (const char *)std::memchr(base + s->pos, '\n', end - s->pos);
                if (!nl) { F.carry.append(base + s->pos, end - s->pos); s->pos = end; break; }
                deliver(base + s->pos, (size_t)(nl - (base + s->pos)));
                s->pos = (size_t)(nl - base) + 1;
            }
            if (s->pos == end) {
                std::lock_guard<std::mutex> lk(mu_);
                F.spare.push_back(std::move(F.segs.front()));
                F.segs.pop_front();
                try_read(f);  // (room in the window: the next fetch; the split itself is busy -- this task)
            }
            if (F.k == R_) break;
        }
        if (at_end && F.k < R_ && !F.carry.empty()) deliver("", 0);  // a last line without its newline
        const double dt = now() - t0;
        std::lock_guard<std::mutex> lk(mu_);
        clock.read += dt; clock.split += dt; ++clock.n_split;
        file_busy_[f] = false;
        if (F.k == R_ || at_end || read_error_) {
            const size_t k = F.k;
            F.k = 0;
            B.got[f] = k;
            file_next_[f] = b + 1;
            if (k < R_) stop_at_ = std::min(stop_at_, b);
            if (++B.reads_done == rd_.size()) start_parse(B);
        }
        try_read(f);
        --in_flight_;
        cv_.notify_all();
    }
    void read_task(size_t f, size_t b) {
        const double t0 = now();
        Block &B = blocks_[b % kBlocks];
        size_t k = 0;
        std::exception_ptr ex;
        try {
            if (have_row_[f]) { B.lines[f][k++] = first_row_[f]; have_row_[f] = 0; }
            while (k < R_ && rd_[f].getline(B.lines[f][k])) ++k;
        } catch (...) { ex = std::current_exception(); }
        const double dt = now() - t0;
        std::lock_guard<std::mutex> lk(mu_);
        clock.read += dt;
        B.got[f] = k;
        if (ex && !read_error_) read_error_ = ex;
        file_busy_[f] = false;
        file_next_[f] = b + 1;
        if (k < R_) stop_at_ = std::min(stop_at_, b);  // this file ends in block b: no position beyond it has a row from every file
        if (++B.reads_done == rd_.size()) start_parse(B);
        try_read(f);
        --in_flight_;
        cv_.notify_all();
    }
    // (mu_ held) every file has delivered its lines of block B: the positions every file still has, cut into chunks
    void start_parse(Block &B) {
        size_t n = R_;
        for (size_t g : B.got) n = std::min(n, g);
        B.n_blk = n;
        const size_t n_parts = n ? (n + chunk_ - 1) / chunk_ : 0;
        B.parts.clear();
        B.parts.resize(std::max<size_t>(n_parts, 1));
        if (read_error_) { B.parts[0].error = read_error_; B.parsed = true; return; }
        if (n_parts == 0) { B.parsed = true; return; }
        for (size_t p = 0; p < n_parts; ++p) {
            ++in_flight_;
            pool_->submit([this, &B, p]() { parse_task(B, p); });
        }
    }
    void parse_task(Block &B, size_t p) {
        const double t0 = now();
        Part &P = B.parts[p];
        const size_t lo = p * chunk_, hi = std::min(B.n_blk, lo + chunk_), NB = rd_.size();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!spare_slabs_.empty()) { P.slab = std::move(spare_slabs_.back()); spare_slabs_.pop_back(); }
        }
        if (!P.slab) P.slab.reset(new SlabBuilder((uint32_t)n_sample_));
        P.slab->reserve_rows(hi - lo);
        P.text.reserve(hi - lo);
        std::vector<std::string> rows(NB);
        try {
            for (size_t r = lo; r < hi; ++r) {
                for (size_t f = 0; f < NB; ++f) rows[f].swap(B.lines[f][r]);
                SiteText st;
                if (parse_site_rows_fast(rows, n_sample_, *P.slab, st)) P.text.push_back(std::move(st));
                for (size_t f = 0; f < NB; ++f) rows[f].swap(B.lines[f][r]);  // (the strings keep their capacity for the next block)
            }
        } catch (...) { P.error = std::current_exception(); }  // the positions before the offending one are in P
        const double dt = now() - t0;
        std::lock_guard<std::mutex> lk(mu_);
        clock.parse += dt;
        if (++B.parts_done == B.parts.size()) B.parsed = true;
        --in_flight_;
        cv_.notify_all();
    }

    std::vector<GzLineReader> &rd_;
    std::vector<std::unique_ptr<BgzfFile>> bg_;   // per file: the BGZF segment pipeline, or null (sequential reader)
    std::vector<std::unique_ptr<SlabBuilder>> spare_slabs_;  // the chunks' builders, reused
    std::vector<std::string> first_row_;
    std::vector<char> have_row_;   // (one byte per file: the files' read tasks touch their own entry concurrently)
    size_t n_sample_;
    int threads_;
    size_t R_ = 0, chunk_ = 1;
    std::vector<Block> blocks_;
    std::vector<size_t> file_next_;
    std::vector<bool> file_busy_;
    size_t freed_ = 0, stop_at_ = (size_t)-1, in_flight_ = 0;
    std::exception_ptr read_error_;
    TaskPool *pool_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_;
};

}  // namespace bvamd
