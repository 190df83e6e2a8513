// producer_check.cpp -- the pipelined batchfile producer (basevar_amd/host/batch_producer.hpp) against the plain loop it
// replaces: one row from every file per position, parsed in order on one thread (src/basetype_caller.cpp:586-611).
//   producer_check a.gz,b.gz,... [expect_error_substring]
// For 1, 2, 3 and 8 producer threads: the same positions in the same order, byte-identical planes and texts; with a malformed
// row somewhere, the same positions delivered before it and the same error text.
#include <cstdio>
#include <iostream>

#include "../../basevar_amd/host/batch_producer.hpp"

struct Out {
    std::vector<uint8_t> cell, phred, mapq, ref;
    std::vector<uint16_t> rank;
    std::vector<std::string> text;
    std::string error;
    bool operator==(const Out &o) const {
        return cell == o.cell && phred == o.phred && mapq == o.mapq && ref == o.ref && rank == o.rank && text == o.text && error == o.error;
    }
};
static void take(Out &o, const bvamd::SlabBuilder &sb, const std::vector<bvamd::SiteText> &text, size_t n) {
    for (size_t i = 0; i < sb.n_sites(); ++i) {
        o.cell.insert(o.cell.end(), sb.cell_row(i), sb.cell_row(i) + n);
        o.phred.insert(o.phred.end(), sb.phred_row(i), sb.phred_row(i) + n);
        o.mapq.insert(o.mapq.end(), sb.mapq_row(i), sb.mapq_row(i) + n);
        o.rank.insert(o.rank.end(), sb.rank_row(i), sb.rank_row(i) + n);
        o.ref.push_back(sb.ref_code(i));
        std::string t = text[i].ref_id + ":" + std::to_string(text[i].ref_pos) + ":" + text[i].ref_base;
        for (const auto &x : text[i].indel_tokens) t += "|" + x;
        o.text.push_back(t);
    }
}
struct Opened {
    std::vector<bvamd::GzLineReader> readers;
    std::vector<std::string> first_row;
    std::vector<bool> have_row;
    std::vector<size_t> header_lines;
    size_t n_sample = 0;
    explicit Opened(const std::vector<std::string> &files) : readers(files.size()), first_row(files.size()), have_row(files.size(), false), header_lines(files.size(), 0) {
        std::vector<std::string> ids;
        for (size_t b = 0; b < files.size(); ++b) {
            if (!readers[b].open(files[b])) throw std::runtime_error("cannot open " + files[b]);
            std::string line;
            while (readers[b].getline(line)) {
                if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); header_lines[b] += line.empty() ? 1 : 0; break; }
                bvamd::parse_sample_ids(line, ids);
                ++header_lines[b];
            }
        }
        n_sample = ids.size();
    }
};

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::vector<std::string> files = bvamd::pieces(argv[1], ',');
    const std::string expect_err = argc > 2 ? argv[2] : "";
    // ---- the plain loop
    Out want;
    size_t n = 0;
    {
        Opened in(files);
        n = in.n_sample;
        std::vector<std::string> rows(files.size());
        bvamd::SlabBuilder sb((uint32_t)n);
        std::vector<bvamd::SiteText> text;
        try {
            for (;;) {
                bool eof = false;
                for (size_t b = 0; b < files.size(); ++b) {
                    if (in.have_row[b]) { rows[b] = in.first_row[b]; in.have_row[b] = false; }
                    else if (!in.readers[b].getline(rows[b])) { eof = true; break; }
                }
                if (eof) break;
                bvamd::SiteText st;
                if (bvamd::parse_site_rows_fast(rows, n, sb, st)) text.push_back(std::move(st));
            }
        } catch (const std::exception &ex) { want.error = ex.what(); }
        take(want, sb, text, n);
    }
    if (!expect_err.empty() && want.error.find(expect_err) == std::string::npos) {
        std::printf("FAIL: the plain loop's error is '%s', expected '%s'\n", want.error.c_str(), expect_err.c_str());
        return 1;
    }
    if (expect_err.empty() && !want.error.empty()) { std::printf("FAIL: unexpected error %s\n", want.error.c_str()); return 1; }
    // ---- the pipeline
    for (int threads : {1, 2, 3, 8}) {
        Out got;
        Opened in(files);
        bvamd::BatchfileProducer producer(in.readers, in.first_row, in.have_row, in.n_sample, threads);
        producer.set_paths(files, in.header_lines);  // (BGZF files then go through the segment pipeline; the others stay sequential)
        try {
            producer.run([&](bvamd::SlabBuilder &part_, std::vector<bvamd::SiteText> &text) {
        bvamd::SlabBuilder *part = &part_;
                take(got, *part, text, n);
                return true;
            });
        } catch (const std::exception &ex) { got.error = ex.what(); }
        if (!(got == want)) {
            std::printf("FAIL: %d threads: %zu positions (error '%s'), the plain loop %zu (error '%s')\n", threads, got.ref.size(), got.error.c_str(),
                        want.ref.size(), want.error.c_str());
            return 1;
        }
    }
    std::printf("OK %zu positions, %zu samples, error '%s'\n", want.ref.size(), n, want.error.c_str());
    return 0;
}
