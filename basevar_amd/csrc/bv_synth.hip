// bv_synth.hip -- device-side synthetic pileup generator (measurement helper, bench only).
//
// Fills slab planes with the synthetic workload of SURVEY.md section 8(d) using a stateless
// counter-based generator: every cell's draws are a pure function of (seed, global site
// index, sample index), so any rank can generate any site range and regenerate it bit for
// bit.  Not part of the reference surface; it never touches the solver.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/basevar_amd_diag.h"

typedef uint32_t bvs_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t bvs_mix(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float bvs_u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

// site classes cycled by (global site index % 20), SURVEY 8(d): 70 % hom-ref, 10 % AF 0.002,
// 10 % AF 0.05, 5 % AF 0.4, 5 % tri-allelic (0.2 / 0.1)
__device__ __forceinline__ void bvs_site_class(uint64_t gsite, float *af1, float *af2) {
    uint32_t c = (uint32_t)(gsite % 20ull);
    *af1 = 0.f; *af2 = 0.f;
    if (c >= 14 && c < 16) *af1 = 0.002f;
    else if (c >= 16 && c < 18) *af1 = 0.05f;
    else if (c == 18) *af1 = 0.4f;
    else if (c == 19) { *af1 = 0.2f; *af2 = 0.1f; }
}

__global__ __launch_bounds__(256) void bv_synth_kernel(bv_synth_params p, uint32_t n_sites, uint32_t n_samples,
                                                       uint64_t pitch, uint8_t *bs, uint8_t *q, uint8_t *mapq,
                                                       uint16_t *rpr, uint8_t *ref_base) {
    const uint32_t site = blockIdx.y;
    if (site >= n_sites) return;
    const uint64_t gsite = p.site_offset + site;
    const uint32_t chunks = (uint32_t)(pitch >> 4);
    // per-site draws
    const uint64_t hs = bvs_mix(p.seed ^ (gsite * 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A5ull);
    const uint32_t ref = (uint32_t)(hs & 3u);
    const uint32_t alt1 = (ref + 1u + (uint32_t)((hs >> 2) % 3u)) & 3u;
    uint32_t alt2 = (ref + 1u + (uint32_t)((hs >> 8) % 3u)) & 3u;
    if (alt2 == alt1) alt2 = (alt2 == ((ref + 1u) & 3u)) ? ((ref + 2u) & 3u) : ((ref + 1u) & 3u);
    float af1, af2;
    bvs_site_class(gsite, &af1, &af2);
    if (blockIdx.x == 0 && threadIdx.x == 0) ref_base[site] = (uint8_t)ref;

    for (uint32_t ch = blockIdx.x * blockDim.x + threadIdx.x; ch < chunks; ch += gridDim.x * blockDim.x) {
        uint32_t wb[4] = {0, 0, 0, 0}, wq[4] = {0, 0, 0, 0}, wm[4] = {0, 0, 0, 0};
        uint32_t wr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t smp = ch * 16u + j;
            uint32_t code = BV_CELL_N, qq = 0, mq = 0, rp = 0;
            if (smp < n_samples) {
                uint64_t h = bvs_mix(hs + (uint64_t)smp * 0xD1342543DE82EF95ull);
                if (bvs_u01(h) < p.coverage) {
                    uint64_t h2 = bvs_mix(h + 0x632BE59BD9B4E019ull);
                    uint64_t h3 = bvs_mix(h2 + 0x9E3779B97F4A7C15ull);
                    // phred ~ clip(round(N(mean, sd)), min, max) by Box-Muller
                    float u1 = fmaxf(bvs_u01(h2), 1e-7f), u2 = bvs_u01(h2 << 24);
                    float g = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
                    int qi = (int)rintf(p.qual_mean + p.qual_sd * g);
                    qi = max((int)p.qual_min, min((int)p.qual_max, qi));
                    qq = (uint32_t)qi;
                    // true allele, then a sequencing error with probability 10^(-q/10)
                    float ua = bvs_u01(h3);
                    uint32_t base = (ua < af1) ? alt1 : ((ua < af1 + af2) ? alt2 : ref);
                    float perr = __expf(-0.2302585093f * (float)qi);
                    if (bvs_u01(h3 << 24) < perr) base = (uint32_t)((h3 >> 3) & 3u);
                    uint32_t strand = (uint32_t)((h >> 7) & 1u);
                    code = base | (strand << 2);
                    uint64_t h4 = bvs_mix(h3 + 0xBF58476D1CE4E5B9ull);
                    if (bvs_u01(h4) < p.indel_frac) code = BV_CELL_INS + (uint32_t)((h4 >> 5) & 1u);
                    mq = (bvs_u01(h4 << 24) < 0.8f) ? 60u : (10u + (uint32_t)((h4 >> 9) % 50u));
                    rp = 1u + (uint32_t)((h4 >> 17) % 100u);
                }
            }
            wb[j >> 2] |= code << (8 * (j & 3));
            wq[j >> 2] |= qq << (8 * (j & 3));
            wm[j >> 2] |= mq << (8 * (j & 3));
            if (p.layout & BV_SLAB_RPR_TAGGED) rp = BV_RPR_TAGGED(code, rp);  // the producer's choice: ranks are <= 100 here
            wr[j >> 1] |= rp << (16 * (j & 1));
        }
        const size_t off = (size_t)site * pitch + (size_t)ch * 16u;
        *reinterpret_cast<bvs_u32x4 *>(bs + off) = bvs_u32x4{wb[0], wb[1], wb[2], wb[3]};
        *reinterpret_cast<bvs_u32x4 *>(q + off) = bvs_u32x4{wq[0], wq[1], wq[2], wq[3]};
        if (mapq) *reinterpret_cast<bvs_u32x4 *>(mapq + off) = bvs_u32x4{wm[0], wm[1], wm[2], wm[3]};
        if (rpr) {
            bvs_u32x4 *r = reinterpret_cast<bvs_u32x4 *>(rpr + off);
            r[0] = bvs_u32x4{wr[0], wr[1], wr[2], wr[3]};
            r[1] = bvs_u32x4{wr[4], wr[5], wr[6], wr[7]};
        }
    }
}

void bv_launch_synth(const bv_synth_params &p, uint32_t n_sites, uint32_t n_samples, uint64_t pitch, uint8_t *bs,
                     uint8_t *q, uint8_t *mapq, uint16_t *rpr, uint8_t *ref_base, hipStream_t stream) {
    uint32_t chunks = (uint32_t)(pitch >> 4);
    uint32_t gx = (chunks + 255u) / 256u;
    if (gx > 64u) gx = 64u;
    // blockIdx.y is limited to 65535: generate in site batches
    for (uint32_t s0 = 0; s0 < n_sites; s0 += 32768u) {
        uint32_t ns = (n_sites - s0 < 32768u) ? (n_sites - s0) : 32768u;
        bv_synth_params pp = p;
        pp.site_offset = p.site_offset + s0;
        hipLaunchKernelGGL(bv_synth_kernel, dim3(gx, ns), dim3(256), 0, stream, pp, ns, n_samples, pitch,
                           bs + (size_t)s0 * pitch, q + (size_t)s0 * pitch, mapq ? mapq + (size_t)s0 * pitch : nullptr,
                           rpr ? rpr + (size_t)s0 * pitch : nullptr, ref_base + s0);
    }
}
