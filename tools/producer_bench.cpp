// producer_bench.cpp -- the host producer of bv_call alone (no GPU): reference-format batchfiles -> slab rows on T threads,
// rows discarded.  Measures what the engine's host can feed it (reference: basetype_caller.cpp:586-611).
//   producer_bench THREADS a.gz,b.gz,...      -> one JSON line: sites, seconds, sites/s, thread-seconds read / parse / join
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
    for (size_t b = 0; b < files.size(); ++b) {
        if (!readers[b].open(files[b])) { std::fprintf(stderr, "cannot open %s\n", files[b].c_str()); return 1; }
        std::string line;
        while (readers[b].getline(line)) {
            if (line.empty() || line[0] != '#') { first_row[b] = line; have_row[b] = !line.empty(); break; }
            bvamd::parse_sample_ids(line, sample_ids);
        }
    }
    const size_t n_sample = sample_ids.size();
    bvamd::BatchfileProducer producer(readers, first_row, have_row, n_sample, threads);
    size_t sites = 0;
    unsigned long long checksum = 0;
    const auto t0 = std::chrono::steady_clock::now();
    producer.run([&](std::unique_ptr<bvamd::SlabBuilder> part, std::vector<bvamd::SiteText> &text) {
        for (size_t i = 0; i < part->n_sites(); ++i) checksum += part->cell_row(i)[(sites + i) % n_sample] * 131u + part->rank_row(i)[(sites + i) * 7 % n_sample] + text[i].ref_pos;
        sites += part->n_sites();
        return true;
    });
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("{\"threads\": %d, \"files\": %zu, \"samples\": %zu, \"sites\": %zu, \"seconds\": %.4f, \"sites_per_s\": %.1f, "
                "\"block_sites\": %zu, \"read_thread_s\": %.3f, \"parse_thread_s\": %.3f, \"join_s\": %.3f, \"checksum\": %llu}\n",
                threads, files.size(), n_sample, sites, dt, sites / dt, producer.block_sites(), producer.clock.read, producer.clock.parse,
                producer.clock.join, checksum);
    return 0;
}
