// bv_call.cpp -- calling phase of `basevar basetype` on the GPU engine: reference-format
// batchfiles in, reference-format VCF + CVG text out.  It is the composition of the rows either
// side of the hot path: batchfile rows -> BatchInfo
// (batchfile.hpp) -> slab -> engine (C ABI) -> VCF/CVG lines (vcf_emit.hpp), following
// _variant_calling_unit (src/basetype_caller.cpp:529-635), with `--pop-group` handled as
// _get_popgroup_info does (src/basetype_caller.cpp:372-410).
//
// Pipeline: the main thread produces batches of sites in genomic order (a slab + one SiteText per site -- not the
// sites' BatchInfo text); `--gpus G` worker threads, one engine on one GPU each, take whichever batch is next; an
// emitter thread writes the results in batch order.  It replaces the reference's fan-out of 100 kb sub-regions over a
// thread pool and its ordered merge of per-task files (_variants_discovery + merge_file_by_line,
// src/basetype_caller.cpp:469-525).
//
//   bv_call --batchfiles a.bf.gz,b.bf.gz --output-vcf out.vcf --output-cvg out.cvg
//           [--pop-group FILE] [--min-af 0.01] [--batch-sites N (default: 2^28 cells / samples, at most 65536)]
//           [--timing FILE.json]
//           [--gpus G] [--devices 0,1,... | --device 0]
//           [--reference ref.fa --contig NAME:LENGTH ...]
//   bv_call -I a.bam [-I b.bam ...] [-L bam.list] -R ref.fa[.gz] --regions CHR:BEG-END[,CHR:BEG-END...] [--mapq 10]
//           [--thread T] ...   (same outputs)
//
// Batchfiles may be bgzip/gzip-compressed or plain (zlib reads all three).  With BAM inputs the pileup
// (pileup.hpp) feeds the engine directly: the same cells the batchfile rows would carry,
// without the text round trip.
#include <algorithm>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <condition_variable>
#include <deque>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <chrono>

#include "basetype_gpu.hpp"
#include "batch_producer.hpp"
#include "batchfile_fast.hpp"
#include "pileup.hpp"
#include "bgzf_tabix.hpp"
#include "vcf_emit.hpp"

namespace {

// One batch of consecutive sites on its way through the pipeline.
struct Batch {
    uint64_t seq = 0;
    bvamd::SlabBuilder slab;
    std::vector<bvamd::SiteText> text;
    bvamd::BaseTypeBatch result;
    std::string error;
    explicit Batch(uint32_t n_samples) : slab(n_samples) {}
};
typedef std::unique_ptr<Batch> BatchPtr;

// Bounded hand-off queue (mutex + condition variables); close() lets the consumers drain and stop.
class BatchQueue {
public:
    explicit BatchQueue(size_t cap) : cap_(cap) {}
    void push(BatchPtr b) {
        std::unique_lock<std::mutex> lk(mu_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(b));
        not_empty_.notify_one();
    }
    BatchPtr pop() {  // nullptr once closed and empty
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return nullptr;
        BatchPtr b = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return b;
    }
    void close() {
        std::lock_guard<std::mutex> lk(mu_);
        closed_ = true;
        not_empty_.notify_all();
    }
private:
    size_t cap_;
    std::deque<BatchPtr> q_;
    bool closed_ = false;
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
};

// Wall-clock seconds per stage, the way the reference prints its phases (src/basetype_caller.cpp:193-215, 625-632).
struct StageClock {
    double read = 0, parse = 0, engine = 0, emit = 0;  // engine: summed over the workers; the others run on one thread each
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// fn(i) for i in [0, n) on up to `threads` threads (contiguous ranges; the caller's thread takes the first)
// An exception in any range (a malformed row, bad_alloc under large batches) is caught on its thread and the first one rethrown
// on the caller's thread once all ranges are done -- the tool's "[ERROR] ..." exit path, not std::terminate.
template <typename Fn>
void parallel_ranges(size_t n, int threads, Fn fn) {
    const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), n));
    if (nt <= 1) { fn(0, 0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs(nt);
    for (size_t t = 1; t < nt; ++t)
        pool.emplace_back([&, t]() { try { fn(t, n * t / nt, n * (t + 1) / nt); } catch (...) { errs[t] = std::current_exception(); } });
    try { fn(0, 0, n / nt); } catch (...) { errs[0] = std::current_exception(); }
    for (auto &th : pool) th.join();
    for (auto &e : errs) if (e) std::rethrow_exception(e);
}

[[noreturn]] void die(const std::string &m) {
    std::cerr << m << std::endl;
    std::exit(1);
}

}  // namespace

int main(int argc, char **argv) {
    std::vector<std::string> batchfiles, bams;
    std::string out_vcf, out_cvg, pop_group_file, reference = ".", regions, bam_list, devices_arg, timing_file;
    int mapq_thd = 10, threads = 4, n_gpus = 1;  // (`-t`: 4, the reference's default, src/basetype_utils.h:33,94)
    std::vector<bvamd::Contig> contigs;
    float user_min_af = 0.01f;  // BaseTypeARGS default, src/basetype_utils.h:94
    uint32_t batch_sites = 0;   // 0 = from a cell budget once the sample count is known
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "--batchfiles") batchfiles = bvamd::pieces(next(), ',');
        else if (a == "--output-vcf") out_vcf = next();
        else if (a == "--output-cvg") out_cvg = next();
        else if (a == "--pop-group") pop_group_file = next();
        else if (a == "--min-af") user_min_af = std::stof(next());
        else if (a == "--batch-sites") batch_sites = (uint32_t)std::stoul(next());
        else if (a == "--timing") timing_file = next();
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--gpus") n_gpus = std::stoi(next());
        else if (a == "--devices") devices_arg = next();
        else if (a == "--reference" || a == "-R") reference = next();
        else if (a == "-I" || a == "--input") bams.push_back(next());
        else if (a == "-L" || a == "--align-file-list") bam_list = next();
        else if (a == "-r" || a == "--regions") regions = next();
        else if (a == "-q" || a == "--mapq") mapq_thd = std::stoi(next());
        else if (a == "-t" || a == "--thread") threads = std::stoi(next());
        else if (a == "--contig") {
            const std::vector<std::string> p = bvamd::pieces(next(), ':');
            if (p.size() != 2) die("--contig wants NAME:LENGTH");
            contigs.push_back({p[0], (uint32_t)std::stoul(p[1])});
        } else die("unknown argument " + a);
    }
    if (!bam_list.empty()) {
        std::ifstream f(bam_list);
        if (!f) die("[ERROR] cannot open " + bam_list);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            const size_t e = line.find_first_of(" \t");
            bams.push_back(e == std::string::npos ? line : line.substr(0, e));
        }
    }
    if (!(user_min_af > 0.f)) die("[ERROR] --min-af must be > 0");  // the reference refuses it too (caller.cpp:73)
    if (n_gpus < 1) die("[ERROR] --gpus must be >= 1");
    // one engine per entry: --devices a,b,... (an ordinal may repeat: several engines on one GPU), else device, device+1, ...
    std::vector<int> devices;
    if (!devices_arg.empty()) {
        const std::vector<std::string> d = bvamd::pieces(devices_arg, ',');
        for (const auto &x : d) devices.push_back(std::stoi(x));
        if (n_gpus != 1 && (size_t)n_gpus != devices.size()) die("[ERROR] --gpus and --devices disagree");
    } else {
        for (int g = 0; g < n_gpus; ++g) devices.push_back(device + g);
    }
    const bool from_bam = !bams.empty();
    if ((batchfiles.empty() && !from_bam) || out_vcf.empty() || out_cvg.empty() || (from_bam && (regions.empty() || reference == ".")))
        die("usage: bv_call (--batchfiles a,b,... | -I a.bam [-I ...] -R ref.fa --regions CHR:BEG-END [--mapq Q]) --output-vcf FILE "
            "--output-cvg FILE [--pop-group FILE] [--min-af F] [--gpus G]");

    // ---- headers: sample ids in batchfile order (caller.cpp:637-665)
    std::vector<bvamd::GzLineReader> readers(batchfiles.size());
    std::vector<std::string> sample_ids;
    std::vector<std::string> first_row(batchfiles.size());
    std::vector<bool> have_row(batchfiles.size(), false);
    try {
        for (const auto &b : bams) sample_ids.push_back(bvamd::BamFile(b, false).sample_name());
    } catch (const std::exception &ex) { die(ex.what()); }
    std::vector<size_t> header_lines(batchfiles.size(), 0);  // lines in front of the first data row
    for (size_t b = 0; b < batchfiles.size() && !from_bam; ++b) {
        if (!readers[b].open(batchfiles[b])) die("[ERROR] " + batchfiles[b] + " open failure.");
        std::string line;
        while (readers[b].getline(line)) {
            if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); header_lines[b] += line.empty() ? 1 : 0; break; }
            bvamd::parse_sample_ids(line, sample_ids);
            ++header_lines[b];
        }
    }
    const size_t n_sample = sample_ids.size();
    if (n_sample == 0) die("[ERROR] no ##SampleIDs= header found in the batchfiles");
    if (from_bam) batchfiles.clear();

    // ---- pop groups (caller.cpp:372-410): sample -> group, later rows override; groups iterate by name
    std::map<std::string, std::vector<size_t>> groups_idx;
    if (!pop_group_file.empty()) {
        std::ifstream in(pop_group_file);
        if (!in) die("[ERROR] Cannot open file: " + pop_group_file);
        std::map<std::string, std::string> sample2group;
        std::string sn, gn, skip;
        while (true) {
            in >> sn >> gn;
            if (in.eof()) break;
            sample2group[sn] = gn;
            std::getline(in, skip, '\n');
        }
        for (size_t i = 0; i < n_sample; ++i) {
            auto it = sample2group.find(sample_ids[i]);
            if (it != sample2group.end()) groups_idx[it->second].push_back(i);
        }
    }
    std::vector<std::string> group_names;
    std::vector<uint8_t> group_id(n_sample, BV_NO_GROUP);
    for (const auto &kv : groups_idx) {
        if (group_names.size() >= BV_MAX_GROUPS) die("[ERROR] more than 255 population groups");
        for (size_t i : kv.second) group_id[i] = (uint8_t)group_names.size();
        group_names.push_back(kv.first);
    }

    // ---- outputs
    // a name that ends in ".gz": BGZF blocks + a tabix index beside it (bgzf_tabix.hpp; caller.cpp:242-254), else plain text
    bvamd::TextOut VCF, CVG;
    try { VCF.open(out_vcf); CVG.open(out_cvg); } catch (const std::exception &ex) { die(ex.what()); }
    std::vector<std::string> add_group_info;
    for (const auto &g : group_names)  // caller.cpp:229-236
        add_group_info.push_back("##INFO=<ID=" + g + "_AF,Number=A,Type=Float,Description=\"Allele frequency in the " + g +
                                 " populations calculated base on LRT, in the range (0,1)\">");
    std::string hv = bvamd::vcf_header(reference, reference, contigs, add_group_info, sample_ids) + "\n";
    std::string hc = bvamd::cvg_header() + "\n";
    VCF.write_header(hv);
    CVG.write_header(hc);

    // ---- batches are bounded by cells (2^28 cells = 5 x 256 MiB of planes per batch in flight), not by a site count that
    // ignores the row length: a launch carries ~0.1 ms of fill and drain whatever its size, so small batches run the engine
    // in its worst regime (round 2's default was 671 sites at 100 k samples: 0.2 of the large-batch rate).  Per pending
    // site the host keeps the slab row and a SiteText, nothing else.
    // The pipeline below holds up to (G + 1) + (2 G + 2) + G + 1 = 4 G + 4 batches at once (queued, on the engines, waiting for the
    // emitter): the cell budget of a batch shrinks with the number of engines so that all of them together stay under 16 GiB of
    // host planes (8 engines: 2^26.4 cells per batch instead of 2^28, 36 x 0.44 GB).
    if (batch_sites == 0) {
        const size_t pitch = (n_sample + 255) / 256 * 256;
        const size_t in_flight = 4 * devices.size() + 4;
        const size_t budget = std::min<size_t>((size_t)1 << 28, (((size_t)16 << 30) / 5) / in_flight);
        const size_t by_cells = budget / std::max<size_t>(pitch, 1);
        batch_sites = (uint32_t)std::min<size_t>(65536, std::max<size_t>(by_cells, 64));
    }

    // ---- the pipeline: producer (this thread) -> G engine workers -> emitter, results written in batch order
    const size_t G = devices.size();
    BatchQueue to_gpu(G + 1), to_emit(2 * G + 2);
    std::mutex err_mu;
    std::string first_error;
    StageClock clk;
    const double t_start = StageClock::now();
    auto fail = [&](const std::string &m) {
        std::lock_guard<std::mutex> g(err_mu);
        if (first_error.empty()) first_error = m;
    };
    std::vector<std::thread> workers;
    for (size_t g = 0; g < G; ++g)
        workers.emplace_back([&, g]() {
            std::unique_ptr<bvamd::BaseTypeEngine> engine;
            // this worker feeds one GPU: keep it (and the staging memory it touches first) on the CPUs of that GPU's NUMA node
            (void)bv_bind_thread_to_device_node(devices[g]);
            try {
                engine.reset(new bvamd::BaseTypeEngine(batch_sites, (uint32_t)n_sample, user_min_af, devices[g]));
            } catch (const std::exception &ex) { fail(ex.what()); }
            for (BatchPtr b; (b = to_gpu.pop());) {
                if (engine) {
                    const double t0 = StageClock::now();
                    // (the producer's choice of layout: short reads -> the rank words carry the calls, basetype_gpu.hpp)
                    try { b->slab.tag_ranks(); b->result = engine->lrt(b->slab); } catch (const std::exception &ex) { b->error = ex.what(); }
                    const double dt = StageClock::now() - t0;
                    std::lock_guard<std::mutex> lk(err_mu);
                    clk.engine += dt;
                } else {
                    b->error = "no engine on device " + std::to_string(devices[g]);
                }
                to_emit.push(std::move(b));
            }
        });
    size_t n_sites = 0, n_variants = 0;
    std::thread emitter([&]() {
        std::map<uint64_t, BatchPtr> waiting;  // finished out of order
        uint64_t next_seq = 0;
        for (BatchPtr b; (b = to_emit.pop());) {
            waiting[b->seq] = std::move(b);
            for (auto it = waiting.find(next_seq); it != waiting.end(); it = waiting.find(next_seq)) {
                Batch &d = *it->second;
                if (!d.error.empty()) fail(d.error);
                else {
                    const double t0 = StageClock::now();
                    // the lines of a batch are formatted by `--thread` threads (ranges of consecutive sites, a text buffer each)
                    // and written in site order
                    const size_t nt = (size_t)std::max(1, threads);
                    std::vector<std::string> cvg_txt(nt), vcf_txt(nt);
                    std::vector<size_t> nv(nt, 0);
                    parallel_ranges(d.text.size(), threads, [&](size_t t, size_t lo, size_t hi) {
                        for (size_t i = lo; i < hi; ++i) {
                            cvg_txt[t] += bvamd::format_cvg_line(d.text[i], d.result.sites[i]);
                            if (d.result.has_variant(i)) {
                                vcf_txt[t] += bvamd::format_vcf_line(d.text[i], d.slab.cell_row(i), d.slab.phred_row(i), n_sample, d.result.sites[i],
                                                                     group_names.empty() ? nullptr : &d.result.group(i, 0), group_names);
                                ++nv[t];
                            }
                        }
                    });
                    try {
                        for (size_t t = 0; t < nt; ++t) {
                            CVG.write_lines(cvg_txt[t]);
                            VCF.write_lines(vcf_txt[t]);
                            n_variants += nv[t];
                        }
                    } catch (const std::exception &ex) { fail(ex.what()); }
                    n_sites += d.text.size();
                    clk.emit += StageClock::now() - t0;
                }
                waiting.erase(it);
                ++next_seq;
            }
        }
    });

    uint64_t seq = 0;
    BatchPtr cur;
    auto fresh = [&]() {
        cur.reset(new Batch((uint32_t)n_sample));
        if (!from_bam) cur->slab.reserve_rows(batch_sites);  // (address space; the pages come as the rows do -- no regrowth copies)
        if (!group_names.empty()) cur->slab.set_groups(group_id, (uint32_t)group_names.size());
        cur->seq = seq++;
    };
    auto ship = [&]() {
        if (cur && cur->slab.n_sites()) to_gpu.push(std::move(cur));
        else if (cur) --seq;
        cur.reset();
    };
    auto still_ok = [&]() { std::lock_guard<std::mutex> g(err_mu); return first_error.empty(); };

    try {
        if (from_bam) {
            // ---- pileup windows -> slab rows, straight from the tile planes (no text round trip); a site is a position that
            // at least one sample covers (the reference skips rows of total depth 0, caller.cpp:718)
            std::vector<std::string> region_list;  // "-r chr:beg-end[,chr:beg-end ...]" (caller.cpp:311-356), in the order given
            region_list = bvamd::pieces(regions, ',');
            std::string fa_seq, fa_of;
            for (const std::string &rg : region_list) {
                const size_t colon = rg.rfind(':'), dash = rg.rfind('-');
                if (colon == std::string::npos || dash == std::string::npos || dash < colon) die("--regions wants CHR:BEG-END[,CHR:BEG-END...]");
                const std::string ref_id = rg.substr(0, colon);
                const uint32_t beg = (uint32_t)std::stoul(rg.substr(colon + 1, dash - colon - 1));
                const uint32_t end = (uint32_t)std::stoul(rg.substr(dash + 1));
                if (fa_of != ref_id) { fa_seq = bvamd::load_fasta_sequence(reference, ref_id); fa_of = ref_id; }
                if (beg < 1 || end < beg || end > fa_seq.size()) die("[ERROR] region outside " + ref_id);
                const double tp0 = StageClock::now();
                bvamd::pileup_region(bams, fa_seq, ref_id, beg, end, mapq_thd, true, threads, [&](const bvamd::PileupTile &t) {
                    size_t next_indel = 0;
                    for (uint32_t pos = t.beg; pos <= t.end && still_ok(); ++pos) {
                        if (t.depth[pos - t.beg] == 0) continue;
                        if (!cur) fresh();
                        const size_t k = t.at(pos, 0);
                        const char rb = fa_seq[pos - 1];
                        cur->slab.add_row(&t.cell[k], &t.qual[k], &t.mapq[k], &t.rank[k],
                                          (uint8_t)bvamd::base_code((char)std::toupper((unsigned char)rb)));
                        bvamd::SiteText st;
                        st.ref_id = ref_id; st.ref_pos = pos; st.ref_base = std::string(1, rb);
                        while (next_indel < t.indels.size() && t.indels[next_indel].pos < pos) ++next_indel;
                        for (; next_indel < t.indels.size() && t.indels[next_indel].pos == pos; ++next_indel)
                            st.indel_tokens.push_back(t.indels[next_indel].text);
                        cur->text.push_back(std::move(st));
                        if (cur->slab.n_sites() == batch_sites) ship();
                    }
                });
                clk.parse += StageClock::now() - tp0;
            }
        } else {
            // ---- one row from every batchfile per position (caller.cpp:586-611), on `--thread` host threads: files read and
            // positions parsed in blocks by a pipeline of tasks (batch_producer.hpp), joined here in position order
            bvamd::BatchfileProducer producer(readers, first_row, have_row, n_sample, threads);
            producer.set_paths(batchfiles, header_lines);  // BGZF files (what the reference writes): members inflated in parallel
            try {
                producer.run([&](bvamd::SlabBuilder &part_, std::vector<bvamd::SiteText> &text) {
                    bvamd::SlabBuilder *part = &part_;
                    size_t done = 0;
                    const size_t have = part->n_sites();
                    while (done < have) {
                        if (!cur) fresh();
                        const size_t room = batch_sites - cur->slab.n_sites(), take = std::min(room, have - done);
                        if (done == 0 && take == have) cur->slab.append(*part);
                        else cur->slab.append_rows(*part, done, take);  // (a part that straddles a batch boundary)
                        for (size_t i = done; i < done + take; ++i) cur->text.push_back(std::move(text[i]));
                        done += take;
                        if (cur->slab.n_sites() == batch_sites) ship();
                    }
                    return still_ok();
                });
            } catch (...) {
                clk.read += producer.clock.read; clk.parse += producer.clock.parse;
                throw;
            }
            clk.read += producer.clock.read; clk.parse += producer.clock.parse;
        }
        ship();
    } catch (const std::exception &ex) { fail(ex.what()); }
    to_gpu.close();
    for (auto &w : workers) w.join();
    to_emit.close();
    emitter.join();
    try { VCF.close(); CVG.close(); } catch (const std::exception &ex) { if (first_error.empty()) first_error = ex.what(); }
    if (!first_error.empty()) die(first_error);
    const double total = StageClock::now() - t_start;
    std::cout << "[INFO] bv_call: " << n_sites << " covered positions, " << n_variants << " VCF records, " << n_sample
              << " samples, " << group_names.size() << " groups, " << G << " engine(s)" << std::endl;
    // stage seconds: read / parse+pack summed over the producer's threads (BAM input: pileup + pack on the producer thread, all under
    // "parse"), engine summed over the workers (staging copies + kernels + records back), emit on the emitter thread; the
    // stages overlap, total is wall time
    char line[512];
    std::snprintf(line, sizeof line,
                  "[INFO] -- %.3f s elapsed, %.1f sites/s: read %.3f s, parse+pack %.3f s (%s; thread-seconds), engine %.3f s (%zu worker(s), batches of %u sites), emit %.3f s",
                  total, total > 0 ? n_sites / total : 0.0, clk.read, clk.parse, from_bam ? "pileup" : "batchfile", clk.engine, G, batch_sites, clk.emit);
    std::cout << line << std::endl;
    if (!timing_file.empty()) {
        std::ofstream tf(timing_file);
        tf << "{\"sites\": " << n_sites << ", \"vcf_records\": " << n_variants << ", \"samples\": " << n_sample << ", \"engines\": " << G
           << ", \"batch_sites\": " << batch_sites << ", \"input\": \"" << (from_bam ? "bam" : "batchfile") << "\", \"parser\": \""
           << (from_bam ? "pileup" : "batchfile") << "\", \"total_s\": " << total << ", \"sites_per_s\": " << (total > 0 ? n_sites / total : 0.0)
           << ", \"read_s\": " << clk.read << ", \"parse_pack_s\": " << clk.parse << ", \"engine_s\": " << clk.engine
           << ", \"emit_s\": " << clk.emit << "}\n";
    }
    return 0;
}
