// Test harness (tests/ only): checks basevar_amd/host/{batchfile,vcf_emit}.hpp
//  (1) the literal reader's tokenisers (oracle/literal_reader.hpp) and the product's formatters against the reference's own ngslib functions (oracle/_ref), when
//      that library is given as argv[2];
//  (2) batchfile writer -> reader round trip;
//  (3) emits CVG/VCF lines for records computed by the oracle restatement (liboracle.so, argv[1])
//      on a small deterministic set of sites, for the Python test to parse and compare.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "../../basevar_amd/host/basetype_gpu.hpp"
#include "../../basevar_amd/host/vcf_emit.hpp"
#include "../../basevar_amd/host/batchfile_fast.hpp"
#include "../../oracle/literal_reader.hpp"  // the literal restatement of the reference's reader: the checker (test infrastructure)

using namespace bvamd;

static int fails = 0;
#define CHECK(cond, msg) do { if (!(cond)) { std::cerr << "FAIL: " << msg << std::endl; ++fails; } } while (0)

typedef int (*split_fn)(const char *, const char *, char *, size_t);
typedef int (*joind_fn)(const double *, int, const char *, char *, size_t);
typedef int (*joini_fn)(const int *, int, const char *, char *, size_t);
typedef int (*joinc_fn)(const char *, int, const char *, char *, size_t);
typedef int (*oracle_run_fn)(const uint8_t *, const uint8_t *, const uint8_t *, const uint16_t *, const uint8_t *,
                             const uint8_t *, uint32_t, uint32_t, uint32_t, uint64_t, double, bv_site_result *,
                             bv_group_result *, int);

static std::string pack(const std::vector<std::string> &v) {
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) { if (i) s.push_back('\x1f'); s += v[i]; }
    return s;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "usage: host_formats_check liboracle.so [libbvref.so]" << std::endl; return 2; }
    void *orc = dlopen(argv[1], RTLD_NOW);
    if (!orc) { std::cerr << dlerror() << std::endl; return 2; }
    oracle_run_fn oracle_run = (oracle_run_fn)dlsym(orc, "oracle_run");

    // ---- (1) primitives vs the reference's
    if (argc > 2) {
        void *ref = dlopen(argv[2], RTLD_NOW);
        if (!ref) { std::cerr << dlerror() << std::endl; return 2; }
        split_fn rs = (split_fn)dlsym(ref, "bvref_split_str"), ri = (split_fn)dlsym(ref, "bvref_split_int"),
                 rc = (split_fn)dlsym(ref, "bvref_split_char");
        joind_fn jd = (joind_fn)dlsym(ref, "bvref_join_double");
        joini_fn ji = (joini_fn)dlsym(ref, "bvref_join_int");
        joinc_fn jc = (joinc_fn)dlsym(ref, "bvref_join_char");
        char buf[4096];
        const char *strs[] = {"chr11\t5246595\tN\t1\t37 0 0\tC N N\tA ! !\t2 0 0\t+ . .", "", "a", "a b", "a  b", " a b ", "+AC N -T",
                              "37 0 0", "12 7", "A ! !", "! ! !", "x\ty", "1,2,,3,", "smp1,smp2", "60 0 13 255"};
        const char *delims[] = {"\t", " ", ","};
        for (const char *s : strs)
            for (const char *d : delims) {
                std::vector<std::string> vs; bvlit::split(std::string(s), vs, d);
                int n = rs(s, d, buf, sizeof buf);
                CHECK(n == (int)vs.size() && pack(vs) == buf, "split<string> on [" << s << "] delim [" << d << "]");
                std::vector<int> vi; bvlit::split(std::string(s), vi, d);
                std::vector<std::string> vis; for (int x : vi) vis.push_back(std::to_string(x));
                n = ri(s, d, buf, sizeof buf);
                CHECK(n == (int)vi.size() && pack(vis) == buf, "split<int> on [" << s << "] delim [" << d << "]: " << pack(vis) << " vs " << buf);
                std::vector<char> vc; bvlit::split(std::string(s), vc, d);
                std::vector<std::string> vcs; for (char x : vc) vcs.push_back(std::to_string((int)x));
                n = rc(s, d, buf, sizeof buf);
                CHECK(n == (int)vc.size() && pack(vcs) == buf, "split<char> on [" << s << "] delim [" << d << "]");
            }
        std::vector<double> dv = {0.0, 1.0, 0.5, 0.500057094119, 1e-7, 1.23456789e-5, 123456.789, 0.1, 1.0 / 3, 2.0 / 3, 1e10, 0.000999999,
                                  0.9999995, 0.99999949, 5e-324, 1e300, -0.0, 3.0e-3, 0.05, 12345678.0, NAN, INFINITY};
        jd(dv.data(), (int)dv.size(), ",", buf, sizeof buf);
        CHECK(join(dv, ",") == buf, "join<double>: " << join(dv, ",") << " vs " << buf);                 // the product's "%g"
        CHECK(bvlit::join(dv, ",") == buf, "literal join<double>");
        std::vector<int> iv = {0, -1, 5, 2147483647, 60};
        ji(iv.data(), (int)iv.size(), " ", buf, sizeof buf);
        CHECK(join(iv, " ") == buf, "join<int>");
        std::vector<char> cv = {'A', '!', 'I', '+', '.', '5'};
        jc(cv.data(), (int)cv.size(), " ", buf, sizeof buf);
        CHECK(join(cv, " ") == buf, "join<char>");
        std::cout << "PRIMITIVES_CHECKED 1" << std::endl;
    }

    // ---- deterministic sites
    const uint32_t N = 90, S = 150, NBF = 3;  // 3 batchfiles of 30 samples each
    uint64_t st = 0x1234567ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    std::vector<BatchInfo> sites;
    for (uint32_t s = 0; s < S; ++s) {
        BatchInfo bi;
        bi.n = N; bi.ref_id = "chr11"; bi.ref_pos = 5246595 + s;
        const char ref = "ACGT"[rnd() & 3];
        const char alt = "ACGT"[(std::string("ACGT").find(ref) + 1 + rnd() % 3) & 3];
        bi.ref_base = std::string(1, (s % 5 == 0) ? (char)std::tolower(ref) : ((s % 11 == 0) ? 'N' : ref));
        const int afpm = (s % 3 == 0) ? 0 : (int)(rnd() % 600);
        for (uint32_t i = 0; i < N; ++i) {
            if (s != 7 && rnd() % 100 < 60) {
                char b = ((int)(rnd() % 1000) < afpm) ? alt : ref;
                if (rnd() % 100 < 2) b = "ACGT"[rnd() & 3];
                int kind = (int)(rnd() % 40);
                bi.align_bases.push_back(kind == 0 ? std::string("+") + b + "T" : (kind == 1 ? std::string("-") + b : std::string(1, b)));
                bi.align_base_quals.push_back((char)(33 + 5 + rnd() % 36));
                bi.mapqs.push_back(rnd() % 5 ? 60 : (int)(rnd() % 60));
                bi.map_strands.push_back(rnd() & 1 ? '+' : '-');
                bi.base_pos_ranks.push_back(1 + (int)(rnd() % 150));
                bi.depth++;
            } else {
                bi.align_bases.push_back("N"); bi.align_base_quals.push_back('!'); bi.mapqs.push_back(0);
                bi.map_strands.push_back('.'); bi.base_pos_ranks.push_back(0);
            }
        }
        sites.push_back(bi);
    }

    // ---- (2) writer -> reader round trip through three batchfile rows per site
    for (const BatchInfo &bi : sites) {
        std::vector<std::string> rows;
        for (uint32_t b = 0; b < NBF; ++b) {
            uint32_t first = b * (N / NBF), cnt = N / NBF, cov = 0;
            for (uint32_t i = first; i < first + cnt; ++i) cov += bi.align_bases[i] != "N";
            std::string row = format_batchfile_row(bi, first, cnt, cov);
            row.pop_back();  // readers strip the newline
            rows.push_back(row);
        }
        BatchInfo back;
        bool ok = bvlit::parse_site_rows(rows, N, back);
        CHECK(ok == (bi.depth > 0), "depth-0 rows are skipped");
        if (ok) {
            CHECK(back.align_bases == bi.align_bases && back.align_base_quals == bi.align_base_quals && back.mapqs == bi.mapqs &&
                  back.map_strands == bi.map_strands && back.base_pos_ranks == bi.base_pos_ranks && back.ref_base == bi.ref_base &&
                  back.ref_pos == bi.ref_pos && back.depth == bi.depth, "batchfile round trip at " << bi.ref_pos);
        }
    }
    {   // malformed rows raise the reference's errors
        bool threw = false;
        try { BatchInfo b; bvlit::parse_site_rows({"chr1\t5\tA\t1\t60"}, 1, b); } catch (const std::runtime_error &e) { threw = std::string(e.what()).find("invalid data") != std::string::npos; }
        CHECK(threw, "short row -> '[ERROR] batchfile has invalid data'");
        threw = false;
        try { BatchInfo b; bvlit::parse_site_rows({"chr1\t5\tA\t1\t60\tA\tI\t3\t+", "chr1\t6\tA\t1\t60\tA\tI\t3\t+"}, 2, b); } catch (const std::runtime_error &e) { threw = std::string(e.what()).find("same genome coordinate") != std::string::npos; }
        CHECK(threw, "coordinate mismatch -> error");
    }

    // ---- (2b) the byte-level reader (batchfile_fast.hpp) against the literal one (batchfile.hpp + SlabBuilder::add_site): the
    // same slab row and SiteText, the same "skipped", or the same exception text -- on the valid rows above and on rows
    // damaged at random (tokens dropped / doubled / emptied, characters replaced, columns removed, signs and junk in numbers)
    {
        struct Outcome {
            int kind = 0;  // 0 row added, 1 skipped (depth 0), 2 threw
            std::string what, planes, text;
        };
        auto snapshot = [&](const SlabBuilder &sb, const SiteText &t) {
            Outcome o;
            const bv_slab sl = sb.slab();
            if (sl.n_sites) {
                o.planes.assign((const char *)sl.base_strand, sl.pitch);
                o.planes.append((const char *)sl.qual, sl.pitch);
                o.planes.append((const char *)sl.mapq, sl.pitch);
                o.planes.append((const char *)sl.rpr, sl.pitch * 2);
                o.planes.push_back((char)sl.ref_base[0]);
            }
            o.text = t.ref_id + "|" + std::to_string(t.ref_pos) + "|" + t.ref_base + "|" + pack(t.indel_tokens);
            return o;
        };
        auto literal = [&](const std::vector<std::string> &rows, size_t n) {
            Outcome o;
            try {
                BatchInfo bi;
                if (!bvlit::parse_site_rows(rows, n, bi)) { o.kind = 1; return o; }
                SlabBuilder sb((uint32_t)n);
                sb.add_site(bi);
                o = snapshot(sb, site_text_of(bi));
            } catch (const std::exception &e) { o.kind = 2; o.what = e.what(); }
            return o;
        };
        auto fast = [&](const std::vector<std::string> &rows, size_t n) {
            Outcome o;
            try {
                SlabBuilder sb((uint32_t)n);
                SiteText t;
                if (!parse_site_rows_fast(rows, n, sb, t)) { o.kind = 1; CHECK(sb.n_sites() == 0 && sb.slab().n_sites == 0, "a skipped row leaves nothing"); return o; }
                o = snapshot(sb, t);
            } catch (const std::exception &e) { o.kind = 2; o.what = e.what(); }
            return o;
        };
        size_t n_valid = 0, n_skipped = 0, n_threw = 0;
        // every case also goes to argv[4] with the FAST reader's outcome: the Python test runs the reference's own
        // _basevar_caller (oracle/_ref/libbvcaller.so) on the same rows and compares outcome kind and exception text
        FILE *dump = argc > 4 ? std::fopen(argv[4], "w") : nullptr;
        auto compare = [&](const std::vector<std::string> &rows, size_t n, const char *tag) {
            const Outcome a = literal(rows, n), b = fast(rows, n);
            if (dump) {
                std::string w = b.what;
                for (char &ch : w) if (ch == '\n') ch = '\x01';
                std::fprintf(dump, "CASE %zu %zu %d %s\n", rows.size(), n, b.kind, w.c_str());
                for (const std::string &r : rows) std::fprintf(dump, "%s\n", r.c_str());
            }
            // (std::stoi's own exception texts are the library's; the two readers call it on the same fields)
            const bool same = a.kind == b.kind && a.what == b.what && a.planes == b.planes && (a.kind != 0 || a.text == b.text);
            CHECK(same, "fast reader != literal reader (" << tag << "): kinds " << a.kind << "/" << b.kind << " [" << a.what << "] vs [" << b.what << "] rows[0]=" << rows[0].substr(0, 200));
            (a.kind == 0 ? n_valid : a.kind == 1 ? n_skipped : n_threw)++;
        };
        for (const BatchInfo &bi : sites) {
            std::vector<std::string> rows;
            for (uint32_t b = 0; b < NBF; ++b) {
                uint32_t first = b * (N / NBF), cnt = N / NBF, cov = 0;
                for (uint32_t i = first; i < first + cnt; ++i) cov += bi.align_bases[i] != "N";
                std::string row = format_batchfile_row(bi, first, cnt, cov);
                row.pop_back();
                rows.push_back(row);
            }
            compare(rows, N, "valid");
            for (int rep = 0; rep < 40; ++rep) {  // damage
                std::vector<std::string> bad = rows;
                const int hits = 1 + (int)(rnd() % 3);
                for (int hgt = 0; hgt < hits; ++hgt) {
                    std::string &r = bad[rnd() % bad.size()];
                    if (r.empty()) continue;
                    const size_t at = rnd() % r.size();
                    switch (rnd() % 9) {
                        case 0: r.erase(at, 1 + rnd() % 3); break;
                        case 1: r.insert(at, " "); break;
                        case 2: r.insert(at, "\t"); break;
                        case 3: r[at] = "ACGTN+-.!x5 \t-"[rnd() % 14]; break;
                        case 4: r.insert(at, "-7"); break;
                        case 5: r.insert(at, "99999999999"); break;
                        case 6: { const size_t sp = r.find(' ', at); if (sp != std::string::npos) r.erase(at, sp - at); break; }
                        case 7: { const size_t tb = r.rfind('\t'); if (tb != std::string::npos && (rnd() & 1)) r.erase(tb); break; }
                        default: r.insert(at, "+ACG"); break;
                    }
                }
                compare(bad, N, "damaged");
            }
        }
        compare({"chr1\t5\tA\t1\t60"}, 1, "short row");
        compare({"chr1\t5\tA\t1\t60\tA\tI\t3\t+", "chr1\t6\tA\t1\t60\tA\tI\t3\t+"}, 2, "coordinate mismatch");
        compare({"chr1\t5\tA\t0\t60\tAC\tI\t3\tx"}, 1, "depth 0 hides a bad token");
        compare({"chr1\t5\tA\t1\t60\tAC\tI\t3\tx"}, 1, "base token of two characters");
        compare({"chr1\t5\tA\t1\t60\tR\tI\t3\t+"}, 1, "base outside ACGT");
        compare({"chr1\t5\tA\t1\t60\tA\tI\t3\tx"}, 1, "strange strand");
        compare({"chr1\t5\tA\t2\t60 \tA N\t I\t3 +4\t+ ."}, 2, "empty tokens");
        // an EMPTY Readbases token: the reference takes its [0] (the terminator), fails size() != 1 and throws (basetype.cpp:50-56)
        compare({"chr1\t5\tA\t1\t60 60\tA \tI I\t3 3\t+ +"}, 2, "empty base token");
        {
            bool threw = false;
            try { SlabBuilder sb(2); SiteText t; parse_site_rows_fast({"chr1\t5\tA\t1\t60 60\tA \tI I\t3 3\t+ +"}, 2, sb, t); }
            catch (const std::runtime_error &e) { threw = std::string(e.what()).find("size of aligned base is not 1") != std::string::npos; }
            CHECK(threw, "an empty base token raises the reference's error");
        }
        compare({"chr1\t5\ta\t1\t-3\t+AT\t\t70000\t-"}, 1, "negative mapq, empty quality, rank past 16 bits");
        // more edges, each also run through the reference's own caller by the Python test (argv[4])
        compare({"chr1\t5\tA\t1\t60\tA\tI\t3\t+\r"}, 1, "CRLF line end");
        compare({"chr1\t5\tA\t1\t60\tA\tI\t3\t+\t"}, 1, "trailing tab");
        compare({"chr1\t99999999999\tA\t1\t60\tA\tI\t3\t+"}, 1, "position past int");
        compare({"chr1\tx5\tA\t1\t60\tA\tI\t3\t+"}, 1, "position not a number");
        compare({"chr1\t5\tA\tone\t60\tA\tI\t3\t+"}, 1, "depth not a number");
        compare({"chr1\t5\tA\t-1\t60\tA\tI\t3\t+", "chr1\t5\tA\t1\t60\tC\tI\t3\t-"}, 2, "depths summing to zero hide the row");
        compare({"chr1\t5\tA\t1\t 60\tA\tI\t3\t+"}, 1, "leading blank in a column");
        compare({"chr1\t5\t\t1\t60\tA\tI\t3\t+"}, 1, "empty reference base");
        compare({"chr1\t5\tAC\t1\t60\tA\tI\t3\t+"}, 1, "two reference bases");
        compare({"chr1\t5\tA\t1\t60\ta\tI\t3\t+"}, 1, "lower-case read base");
        compare({"chr1\t5\tA\t1\t6e1\tA\tI\t3.7\t+"}, 1, "numbers the int reader stops inside");
        compare({"chr1\t5\tA\t2\t60 60\t+A -A\tI I\t3 4\t. ."}, 2, "indels only, no strands");
        compare({"chr1\t5\tA\t2\t60 60\tA N\tI !\t3 0\t+ x"}, 2, "a strange strand on an N call is never looked at");
        if (dump) std::fclose(dump);
        std::cout << "FAST_READER_CASES valid " << n_valid << " skipped " << n_skipped << " threw " << n_threw << std::endl;
        CHECK(n_valid > 100 && n_threw > 100, "the damaged rows exercise both outcomes");
    }

    // ---- (3) records from the oracle restatement -> lines
    SlabBuilder sb(N);
    std::vector<uint8_t> gid(N);
    for (uint32_t i = 0; i < N; ++i) gid[i] = (i % 3 == 0) ? BV_NO_GROUP : (uint8_t)(i % 2);
    sb.set_groups(gid, 2);
    std::vector<const BatchInfo *> kept;
    for (const BatchInfo &bi : sites)
        if (bi.depth > 0) { sb.add_site(bi); kept.push_back(&bi); }
    bv_slab sl = sb.slab();
    std::vector<bv_site_result> rec(sl.n_sites);
    std::vector<bv_group_result> grec((size_t)sl.n_sites * 2);
    oracle_run(sl.base_strand, sl.qual, sl.mapq, sl.rpr, sl.ref_base, sl.group_id, 2, sl.n_sites, sl.n_samples, sl.pitch,
               bv_min_af(N, 0.01f), rec.data(), grec.data(), 1);
    std::vector<std::string> gnames = {"BJ", "GD"};
    std::cout << "CVG_HEADER_BEGIN\n" << cvg_header() << "\nCVG_HEADER_END" << std::endl;
    std::cout << "VCF_HEADER_BEGIN\n" << vcf_header("ref.fa", "/abs/ref.fa", {{"chr11", 135006516}}, {"##INFO=<ID=BJ_AF>"}, {"s1", "s2"}) << "\nVCF_HEADER_END" << std::endl;
    for (size_t i = 0; i < kept.size(); ++i) {
        std::cout << "REC " << i << " " << rec[i].total_depth << " " << (int)rec[i].n_alt << std::endl;
        // the site as three batchfile rows: the Python test hands them to the reference's own _basevar_caller
        // (oracle/_ref/libbvcaller.so) and holds the CVG / VCF lines below against what THAT writes
        for (uint32_t b = 0; b < NBF; ++b) {
            uint32_t first = b * (N / NBF), cnt = N / NBF, cov = 0;
            for (uint32_t k = first; k < first + cnt; ++k) cov += kept[i]->align_bases[k] != "N";
            std::cout << "ROW " << format_batchfile_row(*kept[i], first, cnt, cov);
        }
        std::string c = format_cvg_line(*kept[i], rec[i]);
        if (!c.empty()) std::cout << "CVG " << c;
        std::string v = format_vcf_line(*kept[i], rec[i], &grec[i * 2], gnames);
        if (!v.empty()) std::cout << "VCF " << v;
    }
    FILE *f = std::fopen(argc > 3 ? argv[3] : "/dev/null", "wb");
    std::fwrite(rec.data(), sizeof(bv_site_result), rec.size(), f);
    std::fwrite(grec.data(), sizeof(bv_group_result), grec.size(), f);
    std::fclose(f);
    std::cout << "FAILS " << fails << std::endl;
    return fails ? 1 : 0;
}
