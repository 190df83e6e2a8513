// Test harness (tests/ only): writes a header and data lines through basevar_amd/host/bgzf_tabix.hpp's TextOut into
// argv[1] (a name ending in .gz: BGZF + .tbi).  Lines: `argv[2]` sequences, positions with gaps, some lines longer than a block.
#include <cstdlib>
#include <iostream>

#include "../../basevar_amd/host/bgzf_tabix.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const int n_seq = std::atoi(argv[2]);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    try {
        bvamd::TextOut out;
        out.open(argv[1]);
        out.write_header("##fileformat=TESTv1\n#CHROM\tPOS\tPAYLOAD\n");
        for (int s = 0; s < n_seq; ++s) {
            long pos = 1 + (long)(rnd() % 50000);
            std::string batch;
            for (int i = 0; i < 400; ++i) {
                pos += 1 + (long)(rnd() % ((i % 50 == 0) ? 300000 : 400));  // a few jumps across 16 kb windows and bins
                if (pos > (1l << 29)) break;
                std::string line = "chr" + std::to_string(s + 1) + "\t" + std::to_string(pos) + "\t";
                const size_t len = (i % 97 == 0) ? 150000 : 20 + rnd() % 200;  // some lines span several blocks
                for (size_t k = 0; k < len; ++k) line.push_back((char)('A' + rnd() % 26));
                batch += line + "\n";
                if (i % 7 == 0) { out.write_lines(batch); batch.clear(); }
            }
            out.write_lines(batch);
        }
        out.close();
    } catch (const std::exception &ex) {
        std::cerr << ex.what() << std::endl;
        return 1;
    }
    return 0;
}
