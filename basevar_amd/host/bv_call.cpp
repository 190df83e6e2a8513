// bv_call.cpp -- calling phase of `basevar basetype` on the GPU engine: reference-format
// batchfiles in, reference-format VCF + CVG text out.  It is the composition of the rows either
// side of the hot path: batchfile rows -> BatchInfo
// (batchfile.hpp) -> slab -> engine (C ABI) -> VCF/CVG lines (vcf_emit.hpp), following
// _variant_calling_unit (src/basetype_caller.cpp:529-635), with `--pop-group` handled as
// _get_popgroup_info does (src/basetype_caller.cpp:372-410).
//
//   bv_call --batchfiles a.bf.gz,b.bf.gz --output-vcf out.vcf --output-cvg out.cvg
//           [--pop-group FILE] [--min-af 0.01] [--batch-sites N (default: min(4096, 2^26 / samples))] [--device 0]
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
#include <map>
#include <string>
#include <vector>

#include "basetype_gpu.hpp"
#include "pileup.hpp"
#include "vcf_emit.hpp"

namespace {

struct GzReader {
    gzFile f = nullptr;
    std::string buf;
    bool open(const std::string &path) {
        f = gzopen(path.c_str(), "rb");
        if (f) gzbuffer(f, 1 << 20);
        return f != nullptr;
    }
    bool getline(std::string &line) {
        line.clear();
        char tmp[1 << 16];
        for (;;) {
            if (!gzgets(f, tmp, sizeof tmp)) return !line.empty();
            line += tmp;
            if (!line.empty() && line.back() == '\n') {
                line.pop_back();
                return true;
            }
        }
    }
    ~GzReader() { if (f) gzclose(f); }
};

[[noreturn]] void die(const std::string &m) {
    std::cerr << m << std::endl;
    std::exit(1);
}

}  // namespace

int main(int argc, char **argv) {
    std::vector<std::string> batchfiles, bams;
    std::string out_vcf, out_cvg, pop_group_file, reference = ".", regions, bam_list;
    int mapq_thd = 10, threads = 1;
    std::vector<bvamd::Contig> contigs;
    float user_min_af = 0.01f;  // BaseTypeARGS default, src/basetype_utils.h:94
    uint32_t batch_sites = 0;  // 0 = from a cell budget once the sample count is known
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "--batchfiles") bvamd::split(next(), batchfiles, ",");
        else if (a == "--output-vcf") out_vcf = next();
        else if (a == "--output-cvg") out_cvg = next();
        else if (a == "--pop-group") pop_group_file = next();
        else if (a == "--min-af") user_min_af = std::stof(next());
        else if (a == "--batch-sites") batch_sites = (uint32_t)std::stoul(next());
        else if (a == "--device") device = std::stoi(next());
        else if (a == "--reference" || a == "-R") reference = next();
        else if (a == "-I" || a == "--input") bams.push_back(next());
        else if (a == "-L" || a == "--align-file-list") bam_list = next();
        else if (a == "-r" || a == "--regions") regions = next();
        else if (a == "-q" || a == "--mapq") mapq_thd = std::stoi(next());
        else if (a == "-t" || a == "--thread") threads = std::stoi(next());
        else if (a == "--contig") {
            std::vector<std::string> p; bvamd::split(next(), p, ":");
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
    const bool from_bam = !bams.empty();
    if ((batchfiles.empty() && !from_bam) || out_vcf.empty() || out_cvg.empty() || (from_bam && (regions.empty() || reference == ".")))
        die("usage: bv_call (--batchfiles a,b,... | -I a.bam [-I ...] -R ref.fa --regions CHR:BEG-END [--mapq Q]) --output-vcf FILE "
            "--output-cvg FILE [--pop-group FILE] [--min-af F]");

    // ---- headers: sample ids in batchfile order (caller.cpp:637-665)
    std::vector<GzReader> readers(batchfiles.size());
    std::vector<std::string> sample_ids;
    std::vector<std::string> first_row(batchfiles.size());
    std::vector<bool> have_row(batchfiles.size(), false);
    try {
        for (const auto &b : bams) sample_ids.push_back(bvamd::BamFile(b, false).sample_name());
    } catch (const std::exception &ex) { die(ex.what()); }
    for (size_t b = 0; b < batchfiles.size() && !from_bam; ++b) {
        if (!readers[b].open(batchfiles[b])) die("[ERROR] " + batchfiles[b] + " open failure.");
        std::string line;
        while (readers[b].getline(line)) {
            if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); break; }
            bvamd::parse_sample_ids(line, sample_ids);
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
        if (group_names.size() >= BV_MAX_GROUPS) die("[ERROR] more than 32 population groups");
        for (size_t i : kv.second) group_id[i] = (uint8_t)group_names.size();
        group_names.push_back(kv.first);
    }

    // ---- outputs
    FILE *VCF = std::fopen(out_vcf.c_str(), "w"), *CVG = std::fopen(out_cvg.c_str(), "w");
    if (!VCF) die("[ERROR] " + out_vcf + " open failure.");
    if (!CVG) die("[ERROR] " + out_cvg + " open failure.");
    std::vector<std::string> add_group_info;
    for (const auto &g : group_names)  // caller.cpp:229-236
        add_group_info.push_back("##INFO=<ID=" + g + "_AF,Number=A,Type=Float,Description=\"Allele frequency in the " + g +
                                 " populations calculated base on LRT, in the range (0,1)\">");
    std::string hv = bvamd::vcf_header(reference, reference, contigs, add_group_info, sample_ids) + "\n";
    std::string hc = bvamd::cvg_header() + "\n";
    std::fwrite(hv.data(), 1, hv.size(), VCF);
    std::fwrite(hc.data(), 1, hc.size(), CVG);

    // ---- engine
    // pending sites keep their BatchInfo text (~42 B per cell) next to the 5 B/cell slab until they are emitted:
    // bound a batch by cells (2^26 cells ~ 3 GB of host memory), not by a site count that ignores the row length
    if (batch_sites == 0) {
        const size_t by_cells = ((size_t)1 << 26) / std::max<size_t>(n_sample, 1);
        batch_sites = (uint32_t)std::min<size_t>(4096, std::max<size_t>(by_cells, 1));
    }
    bvamd::BaseTypeEngine engine(batch_sites, (uint32_t)n_sample, user_min_af, device);
    bvamd::SlabBuilder slab((uint32_t)n_sample);
    if (!group_names.empty()) slab.set_groups(group_id, (uint32_t)group_names.size());
    std::vector<bvamd::BatchInfo> pending;
    size_t n_sites = 0, n_variants = 0;

    auto flush = [&]() {
        if (pending.empty()) return;
        bvamd::BaseTypeBatch bt = engine.lrt(slab);
        for (size_t i = 0; i < pending.size(); ++i) {
            std::string c = bvamd::format_cvg_line(pending[i], bt.sites[i]);
            std::fwrite(c.data(), 1, c.size(), CVG);
            if (bt.has_variant(i)) {
                std::string v = bvamd::format_vcf_line(pending[i], bt.sites[i],
                                                       group_names.empty() ? nullptr : &bt.group(i, 0), group_names);
                std::fwrite(v.data(), 1, v.size(), VCF);
                ++n_variants;
            }
        }
        n_sites += pending.size();
        pending.clear();
        slab.clear();
    };

    if (from_bam) {
        // ---- pileup -> BatchInfo -> slab, in the reference's 500 kb steps (caller.cpp:826-846)
        try {
            // "-r chr:beg-end[,chr:beg-end ...]" (caller.cpp:311-356); regions are called in the order given
            std::vector<std::string> region_list;
            bvamd::split(regions, region_list, ",");
            std::string fa_seq, fa_of;
            for (const std::string &rg : region_list) {
                const size_t colon = rg.rfind(':'), dash = rg.rfind('-');
                if (colon == std::string::npos || dash == std::string::npos || dash < colon) die("--regions wants CHR:BEG-END[,CHR:BEG-END...]");
                const std::string ref_id = rg.substr(0, colon);
                const uint32_t beg = (uint32_t)std::stoul(rg.substr(colon + 1, dash - colon - 1));
                const uint32_t end = (uint32_t)std::stoul(rg.substr(dash + 1));
                if (fa_of != ref_id) { fa_seq = bvamd::load_fasta_sequence(reference, ref_id); fa_of = ref_id; }
                if (beg < 1 || end < beg || end > fa_seq.size()) die("[ERROR] region outside " + ref_id);
                for (uint32_t sb = beg; sb < end + 1; sb += 500000u) {
                    const uint32_t se = sb + 500000u - 1 > end ? end : sb + 500000u - 1;
                    bvamd::PosMapVector v;
                    bvamd::fetch_base_in_region(bams, fa_seq, mapq_thd, std::make_tuple(ref_id, sb, se), v, true, threads);
                    for (uint32_t pos = sb; pos <= se; ++pos) {
                        bvamd::BatchInfo bi;
                        if (!bvamd::batchinfo_at(v, fa_seq, ref_id, pos, bi)) continue;
                        slab.add_site(bi);
                        pending.push_back(std::move(bi));
                        if (pending.size() == batch_sites) flush();
                    }
                }
            }
        } catch (const std::exception &ex) { die(ex.what()); }
    }

    // ---- one row from every batchfile per position (caller.cpp:586-611)
    std::vector<std::string> rows(batchfiles.size());
    for (; !from_bam;) {
        bool eof = false;
        for (size_t b = 0; b < batchfiles.size(); ++b) {
            if (have_row[b]) { rows[b] = first_row[b]; have_row[b] = false; }
            else if (!readers[b].getline(rows[b])) { eof = true; break; }
        }
        if (eof) break;
        bvamd::BatchInfo bi;
        if (!bvamd::parse_site_rows(rows, n_sample, bi)) continue;  // total depth 0, caller.cpp:718
        slab.add_site(bi);
        pending.push_back(std::move(bi));
        if (pending.size() == batch_sites) flush();
    }
    flush();
    std::fclose(VCF);
    std::fclose(CVG);
    std::cout << "[INFO] bv_call: " << n_sites << " covered positions, " << n_variants << " VCF records, " << n_sample
              << " samples, " << group_names.size() << " groups" << std::endl;
    return 0;
}
