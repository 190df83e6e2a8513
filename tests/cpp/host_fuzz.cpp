// host_fuzz.cpp -- the reader and the emitter of bv_call WITHOUT an engine, for the sanitizer builds (make -C basevar_amd/csrc
// sanitize) and the damaged-input corpus of tests/test_sanitize_cpu.py:
//   host_fuzz THREADS a.bf[.gz],b.bf[.gz],...
// header scan (sample ids), then every position through the pipelined producer (basevar_amd/host/batch_producer.hpp: read /
// inflate / parse tasks on THREADS threads, reused buffers) into slab rows, and every delivered site through the CVG / VCF
// formatters (basevar_amd/host/vcf_emit.hpp) with a record made up from the row itself -- depths, an ALT for every base seen
// beside the reference, NaN / huge / negative floats -- so that the emitter's string tables and buffers see arbitrary content.
// Whatever the bytes of the input: the run ends with "OK ..." (exit 0) or with ONE message on stderr (exit 1), never a signal.
// Replaces nothing in the reference; the code under test is this repo's replacement of src/basetype_caller.cpp:586-611, 1103-1260.
#include <cmath>
#include <cstdio>
#include <iostream>

#include "../../basevar_amd/host/batch_producer.hpp"
#include "../../basevar_amd/host/vcf_emit.hpp"

int main(int argc, char **argv) {
    if (argc < 3) { std::cerr << "usage: host_fuzz THREADS file,file,..." << std::endl; return 2; }
    const int threads = std::max(1, std::atoi(argv[1]));
    std::vector<std::string> files;
    {
        std::string s = argv[2];
        size_t a = 0;
        while (a <= s.size()) {
            const size_t b = s.find(',', a);
            files.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
    }
    try {
        std::vector<bvamd::GzLineReader> readers(files.size());
        std::vector<std::string> first_row(files.size());
        std::vector<bool> have_row(files.size(), false);
        std::vector<size_t> header_lines(files.size(), 0);
        std::vector<std::string> ids;
        for (size_t b = 0; b < files.size(); ++b) {
            if (!readers[b].open(files[b])) throw std::runtime_error("[ERROR] cannot open " + files[b]);
            std::string line;
            while (readers[b].getline(line)) {
                if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); header_lines[b] += line.empty() ? 1 : 0; break; }
                bvamd::parse_sample_ids(line, ids);
                ++header_lines[b];
            }
        }
        const size_t n = ids.size();
        if (n == 0) throw std::runtime_error("[ERROR] no sample ids in the batchfile headers");
        bvamd::BatchfileProducer prod(readers, first_row, have_row, n, threads);
        prod.set_paths(files, header_lines);
        size_t sites = 0, bytes = 0;
        const std::vector<std::string> group_names = {"g1", "g2"};
        prod.run([&](bvamd::SlabBuilder &part, std::vector<bvamd::SiteText> &text) {
            for (size_t i = 0; i < part.n_sites(); ++i) {
                const uint8_t *cell = part.cell_row(i), *phred = part.phred_row(i);
                bv_site_result r{};
                for (size_t k = 0; k < n; ++k)
                    if (cell[k] < 8) { r.depth[cell[k] & 3] += 1; r.total_depth += 1; r.cvg_sb[(cell[k] >> 2) & 1] += 1; }
                r.status = r.total_depth ? (BV_SITE_COVERED | BV_SITE_VARIANT | BV_SITE_RANKSUM) : 0;
                const int ref = part.ref_code(i);
                for (int b = 0; b < 4; ++b)
                    if (b != ref && r.depth[b] && r.n_alt < BV_MAX_ALT) {
                        r.alt[r.n_alt] = (uint8_t)b;
                        r.af[r.n_alt] = (sites & 1) ? std::nan("") : (double)r.depth[b] / r.total_depth;
                        r.caf[r.n_alt] = 1e-300 * (double)r.depth[b];
                        ++r.n_alt;
                    }
                r.qual = (sites % 3 == 0) ? 10000.0 : (sites % 3 == 1 ? 5000.0 : 1e300);
                r.qd = -0.0; r.cvg_fs = 1e-15; r.cvg_sor = 10000.0; r.var_fs = INFINITY; r.var_sor = 0.333333333333;
                r.mq_ranksum = 1e10; r.rpr_ranksum = -1e10; r.bq_ranksum = std::nan("");
                bv_group_result g[2]{};
                g[0].n_alt = r.n_alt; g[0].total_depth = r.total_depth;
                for (int k = 0; k < r.n_alt; ++k) { g[0].alt[k] = r.alt[k]; g[0].af[k] = 0.5; }
                bytes += bvamd::format_cvg_line(text[i], r).size();
                bytes += bvamd::format_vcf_line(text[i], cell, phred, n, r, g, group_names).size();
                ++sites;
            }
            return true;
        });
        std::printf("OK %zu sites, %zu samples, %zu bytes of text\n", sites, n, bytes);
    } catch (const std::exception &ex) {
        std::cerr << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
