"""Host-side (numpy) synthetic pileup slabs following SURVEY.md section 8(d).

Used by the parity tests to build seeded inputs; the throughput bench uses the
device-side generator ``bv_synth_fill`` (csrc/bv_synth.hip), which draws from the same
distributions with a counter-based RNG.

Cell encoding: see include/basevar_amd.h (bits 0-1 base, bit 2 reverse strand, bit 3 no-call).
"""
import numpy as np

# site classes, cycled by (global site index % 20): 70 % hom-ref, 10 % AF 0.002,
# 10 % AF 0.05, 5 % AF 0.4, 5 % tri-allelic (0.2 / 0.1)
SITE_CLASS_AF = [(0.0, 0.0)] * 14 + [(0.002, 0.0)] * 2 + [(0.05, 0.0)] * 2 + [(0.4, 0.0)] + [(0.2, 0.1)]


def round_up(n, m):
    return (n + m - 1) // m * m


def make_slab(n_sites, n_samples, seed=0, coverage=0.08, indel_frac=0.005, qual_mean=32.0, qual_sd=6.0,
              qual_min=2, qual_max=41, site_offset=0, n_groups=0, pitch=None, class_af=None, ref_n_frac=0.0):
    """Returns a dict of numpy planes (base_strand, qual, mapq, rpr: [S][pitch]; ref_base: [S])."""
    rng = np.random.default_rng(seed)
    S, N = int(n_sites), int(n_samples)
    pitch = int(pitch) if pitch else round_up(N, 16)
    classes = class_af if class_af is not None else SITE_CLASS_AF
    ref = rng.integers(0, 4, size=S, dtype=np.uint8)
    alt1 = (ref + rng.integers(1, 4, size=S, dtype=np.uint8)) % 4
    alt2 = (ref + 1 + ((alt1 - ref) % 4) % 3) % 4  # some base different from ref and alt1
    alt2 = np.where(alt2 == ref, (ref + 2) % 4, alt2).astype(np.uint8)
    alt2 = np.where(alt2 == alt1, (alt1 + 1) % 4, alt2).astype(np.uint8)
    alt2 = np.where(alt2 == ref, (alt2 + 1) % 4, alt2).astype(np.uint8)
    sidx = (np.arange(S) + site_offset) % len(classes)
    af1 = np.array([classes[i][0] for i in sidx])[:, None]
    af2 = np.array([classes[i][1] for i in sidx])[:, None]

    u = rng.random((S, N))
    true_base = np.where(u < af1, alt1[:, None], np.where(u < af1 + af2, alt2[:, None], ref[:, None])).astype(np.uint8)
    q = np.clip(np.rint(rng.normal(qual_mean, qual_sd, size=(S, N))), qual_min, qual_max).astype(np.uint8)
    err = rng.random((S, N)) < np.power(10.0, -q.astype(np.float64) / 10.0)
    base = np.where(err, rng.integers(0, 4, size=(S, N), dtype=np.uint8), true_base).astype(np.uint8)
    covered = rng.random((S, N)) < coverage
    indel = covered & (rng.random((S, N)) < indel_frac)
    strand = rng.integers(0, 2, size=(S, N), dtype=np.uint8)
    code = np.where(covered, base | (strand << 2), 8).astype(np.uint8)
    bs = np.where(indel, 9 + rng.integers(0, 2, size=(S, N), dtype=np.uint8), code).astype(np.uint8)
    mapq = np.where(rng.random((S, N)) < 0.8, 60, rng.integers(10, 60, size=(S, N))).astype(np.uint8)
    rpr = rng.integers(1, 101, size=(S, N)).astype(np.uint16)
    # uncovered cells carry the batchfile's placeholders: qual '!' (0), mapq 0, rank 0, strand '.'
    unc = bs == 8
    q = np.where(unc, 0, q).astype(np.uint8)
    mapq = np.where(unc, 0, mapq).astype(np.uint8)
    rpr = np.where(unc, 0, rpr).astype(np.uint16)
    if ref_n_frac > 0:
        ref = np.where(rng.random(S) < ref_n_frac, 4, ref).astype(np.uint8)

    def pad(a, fill):
        if pitch == N:
            return np.ascontiguousarray(a)
        out = np.full((S, pitch), fill, dtype=a.dtype)
        out[:, :N] = a
        return out

    slab = {
        "n_sites": S, "n_samples": N, "pitch": pitch,
        # padding cells are deliberately garbage-looking (covered 'A', phred 40): the engine
        # must ignore everything at or beyond n_samples
        "base_strand": pad(bs, 0), "qual": pad(q, 40), "mapq": pad(mapq, 60), "rpr": pad(rpr, 7),
        "ref_base": ref, "n_groups": int(n_groups),
    }
    if n_groups:
        # 2 groups holding 30 % / 13 % of samples, the rest ungrouped; more groups
        # split the remainder evenly
        r = rng.random(N)
        gid = np.full(N, 0xFF, dtype=np.uint8)
        edges = [0.30, 0.43] + [0.43 + (0.5 * (k + 1) / max(1, n_groups - 2)) for k in range(max(0, n_groups - 2))]
        lo = 0.0
        for g in range(n_groups):
            hi = edges[g]
            gid[(r >= lo) & (r < hi)] = g
            lo = hi
        slab["group_id"] = gid
    return slab


def tag_ranks(slab):
    """The same slab in the tagged rank layout (BV_SLAB_RPR_TAGGED, include/basevar_amd.h): every word of the rpr plane
    also carries its cell's call, rpr = rank | (call & 3) << 13 | (call >> 3) << 15 -- what a producer does when every
    rank of the slab is <= 8,191.  Cells past n_samples are tagged from their (garbage) padding bytes like any other:
    the engine must ignore them."""
    rp = np.asarray(slab["rpr"], dtype=np.uint16)
    if int(rp[:, :int(slab.get("n_samples", rp.shape[1]))].max(initial=0)) > 0x1FFF:
        raise ValueError("tag_ranks: a read-position rank beyond 8,191 does not fit the tagged layout")
    bs = np.asarray(slab["base_strand"], dtype=np.uint16)
    out = dict(slab)
    out["rpr"] = ((rp & 0x1FFF) | ((bs & 3) << 13) | (((bs >> 3) & 1) << 15)).astype(np.uint16)
    out["layout"] = 1
    return out
