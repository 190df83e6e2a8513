// gen_batchfiles.cpp -- synthetic batchfiles in the reference's format (BaseVarBatchFile_v1.0) for host-pipeline measurements:
//   gen_batchfiles OUT_DIR N_SAMPLES SAMPLES_PER_FILE N_SITES [COVERAGE=0.08] [SEED=1]
// writes OUT_DIR/bf_000.gz ... (BGZF, deflate level 1), each holding SAMPLES_PER_FILE samples of every site (the reference's
// --batch-count), with the cell statistics of SURVEY.md section 8d (coverage, phred ~ N(32, 6), 10 % of the sites carry an ALT).
// g++ -O2 -std=c++17 tools/gen_batchfiles.cpp -lz -o gen_batchfiles
#include <zlib.h>

#include "../basevar_amd/host/bgzf_tabix.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: gen_batchfiles OUT_DIR N_SAMPLES SAMPLES_PER_FILE N_SITES [COVERAGE] [SEED]\n"); return 2; }
    const std::string dir = argv[1];
    const uint32_t n = (uint32_t)std::atoi(argv[2]), per = (uint32_t)std::atoi(argv[3]), sites = (uint32_t)std::atoi(argv[4]);
    const double cov = argc > 5 ? std::atof(argv[5]) : 0.08;
    uint64_t st = argc > 6 ? (uint64_t)std::atoll(argv[6]) * 0x9E3779B97F4A7C15ull + 1 : 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    auto uni = [&]() { return (double)(rnd() >> 11) * 0x1p-53; };
    const uint32_t nf = (n + per - 1) / per;
    std::vector<bvamd::BgzfWriter> out(nf);  // BGZF, as the reference writes its batchfiles (src/basetype_caller.cpp:428)
    for (uint32_t f = 0; f < nf; ++f) {
        char name[64];
        std::snprintf(name, sizeof name, "/bf_%03u.gz", f);
        out[f].open(dir + name, 1);
        std::string h = "##fileformat=BaseVarBatchFile_v1.0\n##SampleIDs=";
        for (uint32_t i = f * per; i < std::min(n, (f + 1) * per); ++i) { if (i != f * per) h += ','; h += "S" + std::to_string(i); }
        h += "\n#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\tReadbases\tReadbasesQuality\tReadPositionRank\tStrand\n";
        out[f].write(h);
    }
    std::string mq, bs, qs, rk, sd, row;
    for (uint32_t s = 0; s < sites; ++s) {
        const char ref = "ACGT"[rnd() & 3];
        const char alt = "ACGT"[(std::string("ACGT").find(ref) + 1 + rnd() % 3) & 3];
        const double af = (s % 10 == 3) ? 0.05 : (s % 10 == 7 ? 0.4 : 0.0);
        for (uint32_t f = 0; f < nf; ++f) {
            mq.clear(); bs.clear(); qs.clear(); rk.clear(); sd.clear();
            uint32_t covered = 0;
            for (uint32_t i = f * per; i < std::min(n, (f + 1) * per); ++i) {
                if (i != f * per) { mq += ' '; bs += ' '; qs += ' '; rk += ' '; sd += ' '; }
                if (uni() < cov) {
                    ++covered;
                    double g = std::sqrt(-2 * std::log(uni() + 1e-300)) * std::cos(6.283185307179586 * uni());
                    int q = (int)std::lround(32 + 6 * g); q = q < 2 ? 2 : (q > 41 ? 41 : q);
                    char b = uni() < af ? alt : ref;
                    if (uni() < std::pow(10.0, -q / 10.0)) b = "ACGT"[rnd() & 3];
                    mq += std::to_string(uni() < 0.8 ? 60 : 10 + (int)(rnd() % 50));
                    if (uni() < 0.005) { bs += (rnd() & 1) ? "+" : "-"; bs += b; bs += "T"; } else bs += b;
                    qs += (char)(33 + q);
                    rk += std::to_string(1 + (int)(rnd() % 100));
                    sd += (rnd() & 1) ? '+' : '-';
                } else { mq += '0'; bs += 'N'; qs += '!'; rk += '0'; sd += '.'; }
            }
            row = "chr1\t" + std::to_string(1000 + s) + "\t" + ref + "\t" + std::to_string(covered) + "\t" + mq + "\t" + bs + "\t" + qs + "\t" + rk + "\t" + sd + "\n";
            out[f].write(row);
        }
    }
    for (auto &f : out) f.close();
    return 0;
}
