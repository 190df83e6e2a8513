// Test harness (tests/ only): the writer half of the pileup against the reference's own.
// The product turns a PileupTile (dense position x sample planes + indel tokens, basevar_amd/host/pileup.hpp) into batchfile rows
// with pileup_rows_text(); the reference turns its per-sample position maps into rows with __write_record_to_batchfile
// (src/basetype_caller.cpp:1027-1101), reached here through oracle/_ref/libbvcaller.so (argv[1]).  Random tiles -- uncovered
// positions, N calls, insertions, deletions, both strands, mapq 0-255, ranks to 65,535 -- are described to both; the text must
// be equal byte for byte.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "../../basevar_amd/host/pileup.hpp"

using namespace bvamd;

typedef char *(*write_rows_fn)(const char *, uint32_t, uint32_t, const char *, size_t, size_t, const uint32_t *, const uint32_t *,
                               const char *const *, const char *const *, const int *, const int *, const char *, const char *, size_t *,
                               char *, size_t);
typedef void (*free_fn)(char *);

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "usage: pileup_rows_check <libbvcaller.so>\n"; return 2; }
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { std::cerr << dlerror() << "\n"; return 2; }
    write_rows_fn ref_rows = (write_rows_fn)dlsym(h, "bvref_write_batchfile_rows");
    free_fn ref_free = (free_fn)dlsym(h, "bvref_caller_free");
    if (!ref_rows || !ref_free) { std::cerr << "symbols missing\n"; return 2; }
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    int fails = 0;
    size_t n_rows = 0, n_cells = 0, n_indels = 0;
    for (int round = 0; round < 60; ++round) {
        const size_t n = 1 + rnd() % 70;
        const uint32_t beg = 1 + (uint32_t)(rnd() % 500), len = 1 + (uint32_t)(rnd() % 40), end = beg + len - 1;
        std::string fa(end + 5, 'A');
        for (char &c : fa) c = "ACGTNacgt"[rnd() % 9];
        PileupTile t;
        t.reset("chr" + std::to_string(1 + round % 22), beg, end, n);
        // the reference's side: one entry per claimed cell
        std::vector<uint32_t> e_sample, e_pos;
        std::vector<std::string> e_ref, e_read;
        std::vector<int> e_mapq, e_rpr;
        std::string e_strand, e_qual;
        const int cover = (int)(rnd() % 100);
        for (uint32_t pos = beg; pos <= end; ++pos) {
            for (size_t i = 0; i < n; ++i) {
                if ((int)(rnd() % 100) >= cover) continue;
                const size_t k = t.at(pos, i);
                const bool rev = rnd() & 1;
                const int kind = (int)(rnd() % 12);  // 0: N call, 1: insertion, 2: deletion, else a base
                const char b = "ACGT"[rnd() & 3];
                t.mapq[k] = (uint8_t)(rnd() % 256);
                t.qual[k] = (uint8_t)(rnd() % 60);
                t.rank[k] = (uint16_t)(1 + rnd() % (rnd() % 8 ? 150 : 65535));
                t.depth[pos - beg]++;
                std::string ref_b(1, b), read_b(1, b);
                if (kind == 0) { t.cell[k] = (uint8_t)(BV_CELL_N | (rev ? BV_CELL_REV : 0)); read_b = "N"; }
                else if (kind == 1) {
                    std::string ins;
                    for (int j = 0, m = 1 + (int)(rnd() % 4); j < m; ++j) ins += "ACGT"[rnd() & 3];
                    t.cell[k] = (uint8_t)(BV_CELL_INS | (rev ? BV_CELL_REV : 0));
                    t.indels.push_back({pos, (uint32_t)i, "+" + read_b + ins});
                    read_b += ins;  // longer than the reference bases: an insertion
                    ++n_indels;
                } else if (kind == 2) {
                    std::string del;
                    for (int j = 0, m = 1 + (int)(rnd() % 4); j < m; ++j) del += "ACGT"[rnd() & 3];
                    t.cell[k] = (uint8_t)(BV_CELL_DEL | (rev ? BV_CELL_REV : 0));
                    t.indels.push_back({pos, (uint32_t)i, "-" + ref_b + del});
                    ref_b += del;   // longer than the read bases: a deletion
                    ++n_indels;
                } else t.cell[k] = (uint8_t)(pileup_base_code(b) | (rev ? BV_CELL_REV : 0));
                e_sample.push_back((uint32_t)i); e_pos.push_back(pos); e_ref.push_back(ref_b); e_read.push_back(read_b);
                e_mapq.push_back((int)t.mapq[k]); e_rpr.push_back((int)t.rank[k]);
                e_strand += rev ? '-' : '+'; e_qual += (char)(t.qual[k] + 33);
                ++n_cells;
            }
        }
        std::string got;
        pileup_rows_text(t, fa, got);
        std::vector<const char *> p_ref, p_read;
        for (size_t k = 0; k < e_ref.size(); ++k) { p_ref.push_back(e_ref[k].c_str()); p_read.push_back(e_read[k].c_str()); }
        size_t rl = 0;
        char err[512] = {0};
        char *exp = ref_rows(t.ref_id.c_str(), beg, end, fa.c_str(), n, e_ref.size(), e_sample.data(), e_pos.data(), p_ref.data(),
                             p_read.data(), e_mapq.data(), e_rpr.data(), e_strand.data(), e_qual.data(), &rl, err, sizeof(err));
        if (!exp) { std::cerr << "reference threw: " << err << "\n"; ++fails; continue; }
        if (got != std::string(exp, rl)) {
            ++fails;
            size_t at = 0;
            while (at < got.size() && at < rl && got[at] == exp[at]) ++at;
            std::cerr << "round " << round << ": rows differ at byte " << at << "\n  product:   " << got.substr(at > 40 ? at - 40 : 0, 120)
                      << "\n  reference: " << std::string(exp, rl).substr(at > 40 ? at - 40 : 0, 120) << "\n";
        }
        n_rows += len;
        ref_free(exp);
    }
    std::cout << "PILEUP_ROWS rounds 60 rows " << n_rows << " claimed cells " << n_cells << " indel tokens " << n_indels << " FAILS " << fails << std::endl;
    return fails ? 1 : 0;
}
