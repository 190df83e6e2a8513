// producer_bench.cpp -- the host producer of bv_call alone (no GPU): reference-format batchfiles -> slab rows on T threads,
// rows discarded.  Measures what the engine's host can feed it (reference: basetype_caller.cpp:586-611).
//   producer_bench THREADS a.gz,b.gz,... [sequential]     -> one JSON line ("sequential": BGZF files through the one-stream reader too): sites, seconds, sites/s, thread-seconds read / parse / join
// g++ -O2 -std=c++17 -pthread -I include tools/producer_bench.cpp -lz -o producer_bench
#include <chrono>
#include <cstdio>
#include <iostream>

#include "../basevar_amd/host/batch_producer.hpp"

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: producer_bench THREADS a.gz,b.gz,...\n"); return 2; }
    const int threads = std::atoi(argv[1]);
    const std::vector<std::string> files = bvamd::pieces(argv[2], ',');
    std::vector<bvamd::GzLineReader> readers(files.size());
    std::vector<std::string> sample_ids, first_row(files.size());
    std::vector<bool> have_row(files.size(), false);
    std::vector<size_t> header_lines(files.size(), 0);
    for (size_t b = 0; b < files.size(); ++b) {
        if (!readers[b].open(files[b])) { std::fprintf(stderr, "cannot open %s\n", files[b].c_str()); return 1; }
        std::string line;
        while (readers[b].getline(line)) {
            if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); header_lines[b] += line.empty() ? 1 : 0; break; }
            bvamd::parse_sample_ids(line, sample_ids);
            ++header_lines[b];
        }
    }
    const size_t n_sample = sample_ids.size();
    bvamd::BatchfileProducer producer(readers, first_row, have_row, n_sample, threads);
    if (!(argc > 3 && std::string(argv[3]) == "sequential")) producer.set_paths(files, header_lines);
    const size_t n_bgzf = producer.bgzf_files();
    size_t sites = 0;
    unsigned long long checksum = 0;
    const auto t0 = std::chrono::steady_clock::now();
    producer.run([&](bvamd::SlabBuilder &part_, std::vector<bvamd::SiteText> &text) {
        bvamd::SlabBuilder *part = &part_;
        for (size_t i = 0; i < part->n_sites(); ++i) checksum += part->cell_row(i)[(sites + i) % n_sample] * 131u + part->rank_row(i)[(sites + i) * 7 % n_sample] + text[i].ref_pos;
        sites += part->n_sites();
        return true;
    });
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"threads\": %d, \"files\": %zu, \"bgzf_files\": %zu, \"samples\": %zu, \"sites\": %zu, \"seconds\": %.4f, \"sites_per_s\": %.1f, "
                "\"block_sites\": %zu, \"read_thread_s\": %.3f, \"parse_thread_s\": %.3f, \"join_s\": %.3f, \"checksum\": %llu}\n",
                threads, files.size(), n_bgzf, n_sample, sites, dt, sites / dt, producer.block_sites(), producer.clock.read, producer.clock.parse,
                producer.clock.join, checksum);
    if (n_bgzf) std::fprintf(stderr, "bgzf: fetch %.3f s / %zu tasks, inflate %.3f s / %zu, split %.3f s / %zu\n", producer.clock.fetch, producer.clock.n_fetch,
                             producer.clock.inflate, producer.clock.n_inflate, producer.clock.split, producer.clock.n_split);
    return 0;
}
