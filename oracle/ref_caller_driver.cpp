// TEST INFRASTRUCTURE (oracle/): the reference's own per-position caller, text in -> text out.
//
// _ref/libbvcaller.so = /root/reference/src/basetype_caller.cpp + basetype.cpp + utils.cpp + basetype_utils.cpp +
// htslib/kfunc.c, compiled where they lie (recipe: oracle/Makefile), plus this file.  It exports one function that hands a
// position's batchfile lines to the reference's `_basevar_caller` (src/basetype_caller.cpp:667-762: the reference's reader
// ngslib::split, BaseType + lrt, __gb per pop-group, _out_cvg_line and _out_vcf_line) and returns the bytes that function
// writes.  With it tests/ hold the product's text layers (batchfile rows -> slab, f3: VCF / CVG lines) against the
// reference's OBJECT CODE instead of against a restatement of it.
//
// What is NOT the reference here, and why this is not a build of the reference binary: `_basevar_caller` writes through
// htslib's bgzf_write(BGZF *, ...).  htslib is not built in this image (its generated config.h / version.h are missing, the
// reference's build system is not run), so the library is linked with htslib's symbols UNDEFINED -- nothing on this path calls
// any of them but bgzf_write -- and bgzf_write is defined below as a capture of the bytes it is handed (the `BGZF *` it gets is
// this file's own buffer).  Everything upstream of that call is the reference's code, unmodified; compression, file headers,
// bgzip framing and the tabix index are not exercised and stay unpinned.
//
// Linking: the reference's translation units are compiled with -fvisibility=hidden -ffunction-sections and the library is
// linked with --gc-sections, so only what `_basevar_caller` and `cvg_header_define` reach is kept: the BAM / FASTA / tabix
// paths of basetype_caller.cpp, which take the ADDRESS of htslib functions, are dropped with their references.
//
// Never used by the product (tests/test_abi_cpu.py::test_product_never_touches_the_oracle).
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "basetype_caller.h"  // the reference's own header

namespace {
struct Sink { std::string bytes; };
}

// htslib/bgzf.h:196 -- the one htslib function on the path: captured, not compressed
extern "C" ssize_t bgzf_write(BGZF *fp, const void *data, size_t length) {
    reinterpret_cast<Sink *>(fp)->bytes.append(static_cast<const char *>(data), length);
    return (ssize_t)length;
}

#define BVREF_EXPORT __attribute__((visibility("default")))
extern "C" {

// lines[n_lines]: one row of every batchfile for ONE position.  Pop-groups: names[n_groups], and for group g the sample
// indices idx[off[g] .. off[g + 1]).  Returns 0 (no variant), 1 (variant), -1 (the reference threw: its message in err);
// *vcf / *cvg are malloc'ed copies of what the reference wrote (the caller frees them with bvref_caller_free).
BVREF_EXPORT int bvref_caller_position(const char *const *lines, int n_lines, const char *const *names, const size_t *off, const size_t *idx,
                          int n_groups, double min_af, size_t n_sample, char **vcf, size_t *vcf_len, char **cvg, size_t *cvg_len,
                          char *err, size_t err_cap) {
    *vcf = *cvg = nullptr;
    *vcf_len = *cvg_len = 0;
    try {
        std::vector<std::string> rows;
        for (int i = 0; i < n_lines; ++i) rows.emplace_back(lines[i]);
        std::map<std::string, std::vector<size_t>> groups;
        for (int g = 0; g < n_groups; ++g) groups[names[g]] = std::vector<size_t>(idx + off[g], idx + off[g + 1]);
        Sink v, c;
        const bool variant = _basevar_caller(rows, groups, min_af, n_sample, reinterpret_cast<BGZF *>(&v), reinterpret_cast<BGZF *>(&c));
        *vcf = (char *)malloc(v.bytes.size() + 1); std::memcpy(*vcf, v.bytes.c_str(), v.bytes.size() + 1); *vcf_len = v.bytes.size();
        *cvg = (char *)malloc(c.bytes.size() + 1); std::memcpy(*cvg, c.bytes.c_str(), c.bytes.size() + 1); *cvg_len = c.bytes.size();
        return variant ? 1 : 0;
    } catch (const std::exception &e) {
        if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = '\0'; }
        return -1;
    }
}

BVREF_EXPORT void bvref_caller_free(char *p) { free(p); }

// The writer half of the reference's pileup: `__write_record_to_batchfile` (src/basetype_caller.cpp:1027-1101) turns
// the per-sample position maps that its CIGAR walk filled into batchfile rows.  Entries: (sample, position, reference bases, read
// bases, mapq, read-position rank, strand, quality character); returns the rows it writes for [beg, end] (malloc'ed).  The CIGAR
// walk itself (`__seek_position`) reads htslib's bam1_t through htslib functions and stays out of this library.
BVREF_EXPORT char *bvref_write_batchfile_rows(const char *ref_id, uint32_t beg, uint32_t end, const char *fa_seq, size_t n_samples,
                                              size_t n_entries, const uint32_t *e_sample, const uint32_t *e_pos,
                                              const char *const *e_ref_base, const char *const *e_read_base, const int *e_mapq,
                                              const int *e_rpr, const char *e_strand, const char *e_qual, size_t *len, char *err,
                                              size_t err_cap) {
    try {
        PosMapVector v(n_samples);
        for (size_t k = 0; k < n_entries; ++k) {
            AlignBaseInfo a;
            a.ref_id = ref_id; a.ref_pos = e_pos[k]; a.ref_base = e_ref_base[k]; a.read_base = e_read_base[k];
            a.mapq = e_mapq[k]; a.rpr = e_rpr[k]; a.map_strand = e_strand[k]; a.read_base_qual = e_qual[k];
            v[e_sample[k]][e_pos[k]] = a;
        }
        Sink out;
        __write_record_to_batchfile(v, std::string(fa_seq), std::make_tuple(std::string(ref_id), beg, end), reinterpret_cast<BGZF *>(&out));
        char *p = (char *)malloc(out.bytes.size() + 1);
        std::memcpy(p, out.bytes.c_str(), out.bytes.size() + 1);
        *len = out.bytes.size();
        return p;
    } catch (const std::exception &e) {
        if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = '\0'; }
        return nullptr;
    }
}

// the CVG file's header lines as the reference defines them (src/basetype_utils.cpp:73-88; pure string work)
BVREF_EXPORT char *bvref_cvg_header(void) {
    const std::string h = cvg_header_define(std::vector<std::string>(), std::vector<char>{'A', 'C', 'G', 'T'});
    char *p = (char *)malloc(h.size() + 1);
    std::memcpy(p, h.c_str(), h.size() + 1);
    return p;
}
}
