// bv_pileup.cpp -- batchfile creation phase of `basevar basetype` (_create_batchfiles ->
// __create_a_batchfile, src/basetype_caller.cpp:412-470, 800-874) without htslib: BAM files of one
// batch of samples + the reference FASTA + a region -> one reference-format batchfile, which
// `bv_call --batchfiles` (or the reference's own calling phase) consumes.  Host-only: no GPU involved.
//
//   bv_pileup -R ref.fa[.gz] --regions CHR:BEG-END [--mapq 10] -I a.bam [-I b.bam ...] [-L bam.list]
//             [--filename-has-samplename] [--no-index] [--thread T] [--window POSITIONS] -o out.bf[.gz]
//
// Sample ids come from the first @RG SM tag of each BAM (BamHeader::get_sample_name) or, with
// --filename-has-samplename, from the file name up to its first '.' (src/basetype_caller.cpp:262-294).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "bgzf_tabix.hpp"
#include "pileup.hpp"

namespace {
[[noreturn]] void die(const std::string &m) {
    std::cerr << m << std::endl;
    std::exit(1);
}
std::string basename_of(const std::string &p) {
    const size_t s = p.find_last_of('/');
    return s == std::string::npos ? p : p.substr(s + 1);
}
}  // namespace

int main(int argc, char **argv) {
    std::vector<std::string> bams;
    std::string fasta, regions, out_path, bam_list;
    int mapq = 10;  // BaseTypeARGS default, src/basetype_utils.h
    int threads = 1;
    uint32_t window = 0;  // positions per pileup window (0 = from a cell budget); the rows do not depend on it
    bool name_from_file = false, use_index = true;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "-I" || a == "--input") bams.push_back(next());
        else if (a == "-L" || a == "--align-file-list") bam_list = next();
        else if (a == "-R" || a == "--reference") fasta = next();
        else if (a == "-r" || a == "--regions") regions = next();
        else if (a == "-q" || a == "--mapq") mapq = std::stoi(next());
        else if (a == "-o" || a == "--output") out_path = next();
        else if (a == "--filename-has-samplename") name_from_file = true;
        else if (a == "--no-index") use_index = false;
        else if (a == "-t" || a == "--thread") threads = std::stoi(next());
        else if (a == "--window") window = (uint32_t)std::stoul(next());
        else die("unknown argument " + a);
    }
    if (!bam_list.empty()) {  // get_firstcolumn_from_file, src/basetype_caller.cpp:114-117
        std::ifstream f(bam_list);
        if (!f) die("[ERROR] cannot open " + bam_list);
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            const size_t e = line.find_first_of(" \t");
            bams.push_back(e == std::string::npos ? line : line.substr(0, e));
        }
    }
    if (bams.empty() || fasta.empty() || regions.empty() || out_path.empty())
        die("usage: bv_pileup -R ref.fa --regions CHR:BEG-END [--mapq Q] -I a.bam [-I ...] [-L list] -o out.bf[.gz]");
    try {
        // "chr:beg-end", 1-based inclusive (the chromosome name may itself contain ':')
        const size_t colon = regions.rfind(':'), dash = regions.rfind('-');
        if (colon == std::string::npos || dash == std::string::npos || dash < colon) die("--regions wants CHR:BEG-END");
        const std::string ref_id = regions.substr(0, colon);
        const uint32_t beg = (uint32_t)std::stoul(regions.substr(colon + 1, dash - colon - 1));
        const uint32_t end = (uint32_t)std::stoul(regions.substr(dash + 1));
        if (beg < 1 || end < beg) die("--regions wants 1 <= BEG <= END");

        std::vector<std::string> sample_ids;
        for (const auto &b : bams) {
            if (name_from_file) {
                std::string fn = basename_of(b);
                const size_t ext = fn.rfind('.');  // remove_filename_extension
                if (ext != std::string::npos && ext > 0) fn = fn.substr(0, ext);
                const size_t si = fn.find('.');
                sample_ids.push_back(si > 0 && si != std::string::npos ? fn.substr(0, si) : fn);
            } else {
                sample_ids.push_back(bvamd::BamFile(b, false).sample_name());
            }
        }
        const std::string fa_seq = bvamd::load_fasta_sequence(fasta, ref_id);
        if (end > fa_seq.size()) die("[ERROR] region end beyond the end of " + ref_id);

        const bool gz = out_path.size() > 3 && out_path.compare(out_path.size() - 3, 3, ".gz") == 0;
        // `.gz`: BGZF, as the reference writes its batchfiles ("must be compressed by BGZF", src/basetype_caller.cpp:428) -- a chain
        // of independent gzip members, which bv_call inflates in parallel (batch_producer.hpp); gzip tools read it as they read .gz
        bvamd::BgzfWriter zf;
        std::FILE *pf = nullptr;
        if (gz) zf.open(out_path);
        else { pf = std::fopen(out_path.c_str(), "wb"); if (!pf) die("[ERROR] " + out_path + " open failure."); }
        auto sink = [&](const std::string &s) {
            if (s.empty()) return;
            if (gz) zf.write(s);
            else if (std::fwrite(s.data(), 1, s.size(), pf) != s.size()) throw std::runtime_error("[ERROR] fail to write data");
        };
        const bool has_data = bvamd::create_a_batchfile(bams, sample_ids, fa_seq, std::make_tuple(ref_id, beg, end), mapq, sink, use_index, threads, window);
        if (gz) zf.close(); else std::fclose(pf);
        std::cerr << "[INFO] " << out_path << ": " << bams.size() << " samples, " << ref_id << ":" << beg << "-" << end
                  << (has_data ? "" : " (no covering reads)") << std::endl;
    } catch (const std::exception &ex) {
        die(ex.what());
    }
    return 0;
}
