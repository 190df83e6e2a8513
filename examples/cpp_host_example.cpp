// cpp_host_example.cpp -- the reference-side calling pattern over the C++ wrapper.
// Builds BatchInfo-shaped sites (same members as src/basetype.h:25-43), runs them through the
// engine in one batch, prints a VCF-INFO-like line per variant site, and (argv[1], optional)
// dumps the packed planes + records so that tests can check them against the oracle.
//
//   g++ -std=c++17 -Iinclude examples/cpp_host_example.cpp -Lbasevar_amd/lib -lbasevar_amd
//       (then) -Wl,-rpath,$PWD/basevar_amd/lib -o /tmp/cpp_host_example; /tmp/cpp_host_example /tmp/dump.bin
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../basevar_amd/host/basetype_gpu.hpp"

struct BatchInfo {  // member-for-member the reference's struct
    size_t n = 0;
    std::string ref_id, ref_base;
    uint32_t ref_pos = 0, depth = 0;
    std::vector<std::string> align_bases;
    std::vector<char> align_base_quals;
    std::vector<int> mapqs;
    std::vector<char> map_strands;
    std::vector<int> base_pos_ranks;
};

int main(int argc, char **argv) {
    const uint32_t N = 3000, S = 40;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    bvamd::SlabBuilder builder(N);
    for (uint32_t s = 0; s < S; ++s) {
        BatchInfo bi;
        bi.n = N; bi.ref_id = "chr1"; bi.ref_pos = 1000 + s;
        const char ref = "ACGT"[rnd() & 3];
        const char alt = "ACGT"[(std::string("ACGT").find(ref) + 1 + rnd() % 3) & 3];
        bi.ref_base = std::string(1, (s % 7 == 0) ? (char)std::tolower(ref) : ref);
        const double af = (s % 4 == 0) ? 0.0 : 0.02 * (s % 9);
        for (uint32_t i = 0; i < N; ++i) {
            if (rnd() % 100 < 15) {
                char b = ((rnd() % 10000) < af * 10000) ? alt : ref;
                if (rnd() % 1000 < 3) b = "ACGT"[rnd() & 3];
                bi.align_bases.push_back(rnd() % 300 == 0 ? std::string("+AC") : std::string(1, b));
                bi.align_base_quals.push_back((char)(33 + 20 + rnd() % 20));
                bi.mapqs.push_back(rnd() % 5 ? 60 : (int)(rnd() % 60));
                bi.map_strands.push_back(rnd() & 1 ? '+' : '-');
                bi.base_pos_ranks.push_back(1 + (int)(rnd() % 100));
                bi.depth++;
            } else {
                bi.align_bases.push_back("N");
                bi.align_base_quals.push_back('!');
                bi.mapqs.push_back(0);
                bi.map_strands.push_back('.');
                bi.base_pos_ranks.push_back(0);
            }
        }
        builder.add_site(bi);
    }
    bvamd::BaseTypeEngine engine(S, N, 0.01f, 0);
    bvamd::BaseTypeBatch bt = engine.lrt(builder);
    for (uint32_t s = 0; s < S; ++s) {
        if (!bt.has_variant(s)) continue;
        std::string alts;
        for (char b : bt.get_alt_bases(s)) alts += b;
        bvamd::StrandBiasInfo sb = bt.strand_bias(s, true);
        std::printf("site %u ALT=%s QUAL=%f CM_DP=%d CM_AF=%g MQRankSum=%d ReadPosRankSum=%d BaseQRankSum=%d FS=%f SOR=%f\n", s,
                    alts.c_str(), bt.get_var_qual(s), bt.get_total_depth(s), bt.get_lrt_af(s, alts[0]), bt.mq_rank_sum(s),
                    bt.read_pos_rank_sum(s), bt.base_q_rank_sum(s), sb.fs, sb.sor);
    }
    if (argc > 1) {
        bv_slab sl = builder.slab();
        FILE *f = std::fopen(argv[1], "wb");
        uint64_t hdr[3] = {sl.n_sites, sl.n_samples, sl.pitch};
        std::fwrite(hdr, sizeof hdr, 1, f);
        std::fwrite(sl.base_strand, 1, (size_t)S * sl.pitch, f);
        std::fwrite(sl.qual, 1, (size_t)S * sl.pitch, f);
        std::fwrite(sl.mapq, 1, (size_t)S * sl.pitch, f);
        std::fwrite(sl.rpr, 2, (size_t)S * sl.pitch, f);
        std::fwrite(sl.ref_base, 1, S, f);
        std::fwrite(bt.sites.data(), sizeof(bv_site_result), S, f);
        std::fclose(f);
    }
    // the producer's choice of layout: every rank here is a short read's, so the rank words can carry the calls
    // (BV_SLAB_RPR_TAGGED, include/basevar_amd.h) -- the records must not change by a bit
    if (builder.tag_ranks()) {
        bvamd::BaseTypeBatch bt2 = engine.lrt(builder);
        const bool same = std::memcmp(bt2.sites.data(), bt.sites.data(), sizeof(bv_site_result) * S) == 0;
        std::printf("tagged rank layout: records %s\n", same ? "identical" : "DIFFER");
        if (!same) return 1;
    }
    return 0;
}
