"""GPU tests (-m gpu) of the tagged rank layout (BV_SLAB_RPR_TAGGED, include/basevar_amd.h): the producer stores every cell's
call in the three bits a read-position rank <= 8,191 leaves free, so that the rank sums of a variant site
(ref_vs_alt_ranksumtest, src/basetype.cpp:201-242 via basetype_caller.cpp:1151-1154) read mapq + rpr only -- SURVEY 8d's
3 bytes per cell.  The bar: every record BYTE-IDENTICAL to the one the plain layout gives (which test_gpu_parity.py holds to
the reference), through every kernel that reads the rank plane: long rows, the fused short-row kernel, the LDS-DMA and the
plain pass-2 kernels, the pop-group kernels, chained launches, and tile jobs in both realisations."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from basevar_amd.synth import make_slab, tag_ranks
from test_gpu_parity import bv, check, oracle_run, run_engine  # noqa: F401  (bv: the module fixture)


def both(bv, slab, maf, flags=0):
    out = []
    for s in (slab, tag_ranks(slab)):
        eng = bv.BaseTypeEngine(max_sites=slab["base_strand"].shape[0], min_af_value=maf, device=0, flags=flags)
        try:
            out.append(eng.lrt(s))
        finally:
            eng.close()
    return out


def same(a, b):
    assert a.sites.tobytes() == b.sites.tobytes(), [f for f in a.sites.dtype.names if not np.array_equal(a.sites[f], b.sites[f], equal_nan=a.sites[f].dtype.kind == "f")]
    if a.groups is not None:
        assert a.groups.tobytes() == b.groups.tobytes()
    assert a.n_variant == b.n_variant


@pytest.mark.parametrize("n,sites,groups,flags", [
    (100000, 96, 0, 0),       # the headline shape: bv_pass2_kernel<256, true, false, .., TAG>
    (100003, 40, 0, 0),       # ragged tail
    (70001, 64, 2, 0),        # long rows with pop-groups: the call plane is read for the groups, the rank words masked
    (60000, 48, 14, 0),       # ... 16-bit group counters
    (10000, 700, 0, 0),       # configs[1]: the fused kernel's pass-2 rows as mapq + ranks
    (10007, 300, 0, 0),       # ragged
    (4097, 300, 0, 0), (6143, 200, 0, 0), (6145, 200, 0, 0), (49151, 64, 0, 0),   # slot boundaries of the fused kernel
    (10000, 400, 0, 1 << 16),  # one workgroup: every queue overflows
    (12000, 500, 2, 0),       # fused kernel + streaming group tallies
    (12000, 500, 9, 0),       # more than seven groups: workgroup-per-row pass 2 behind the fused kernel
    (12000, 500, 0, 0x9000),  # the three launches: bv_pass2_dma_kernel<TAG>
    (12003, 300, 0, 0x9000),
    (12000, 500, 0, 0xA000),  # the fused kernel for pass 1, bv_pass2_dma_kernel<TAG> behind it
    (12000, 300, 0, 0x9020),  # plain-load pass-2 kernels (window sweeps)
    (3000, 600, 0, 0),        # rows of <= 4,096 samples: bv_pass2_dma_kernel
    (1500, 600, 2, 0),        # wave per row
    (300, 900, 3, 0), (17, 200, 0, 0), (1, 64, 0, 0),
], ids=lambda v: str(v))
def test_tagged_ranks_give_the_plain_layouts_records(bv, n, sites, groups, flags):
    slab = make_slab(sites, n, seed=1000 + n % 97 + groups, coverage=0.1 if n > 400 else 0.6, n_groups=groups, ref_n_frac=0.03, site_offset=5)
    maf = bv.min_af(n)
    plain, tagged = both(bv, slab, maf, flags)
    same(plain, tagged)
    assert plain.n_variant >= (2 if n > 16 else 0)
    rs = (plain.sites["status"] & 0x10) != 0
    assert rs.sum() == plain.n_variant  # the rank sums were formed for every variant site


def test_tagged_ranks_against_the_oracle(bv, restatement):
    """... and, for one long-row and one short-row shape, directly against the oracle (fed the PLAIN ranks)."""
    for n, sites in ((100000, 64), (10000, 400)):
        slab = make_slab(sites, n, seed=77 + n, coverage=0.08, n_groups=0, ref_n_frac=0.03)
        maf = bv.min_af(n)
        got = run_engine(bv, tag_ranks(slab), maf)
        exp, gexp, margins = oracle_run(restatement, slab, maf)
        check(got, exp, gexp, margins)


@pytest.mark.parametrize("n,groups,flags", [(3000, 0, 0), (3000, 2, 0), (9000, 0, 0), (9000, 0, 0x9000), (20000, 0, 0), (70000, 0, 0), (70000, 2, 0)],
                         ids=["dma_rows", "wave_per_row_groups", "fused_rows", "three_launches", "fused_rows_20k", "long_rows", "long_rows_groups"])
def test_tagged_long_reads_take_the_window_sweeps(bv, n, groups, flags):
    """Ranks of 256 .. 8,191 (long reads) do not fit the perm form's 256-rank window: such a row is re-done by the window
    sweeps, which must take the rank and leave the tag (a tagged word is >= 8,192 whenever its cell is not an 'A')."""
    slab = make_slab(40, n, seed=43, coverage=0.5, class_af=[(0.3, 0.0), (0.2, 0.1)], n_groups=groups)
    rng = np.random.default_rng(5)
    cov = slab["base_strand"] < 8
    slab["rpr"] = np.where(cov, rng.integers(1, 5000, size=slab["rpr"].shape), 0).astype(np.uint16)
    slab["rpr"][3, :] = np.where(cov[3], 8191 - (np.arange(slab["rpr"].shape[1]) % 7), 0)
    slab["rpr"][5, :] = np.where(cov[5], 255, 0)   # the window's last rank: stays in the perm form
    slab["rpr"][6, :] = np.where(cov[6], 256, 0)   # the first one beyond it
    slab["rpr"][8:20, :] = np.where(cov[8:20], rng.integers(1, 150, size=slab["rpr"][8:20].shape), 0)  # short reads among them
    maf = bv.min_af(n)
    plain, tagged = both(bv, slab, maf, flags)
    same(plain, tagged)
    assert plain.n_variant >= 20


def test_ranks_beyond_the_tag_are_the_producers_business(bv):
    """tag_ranks refuses a slab with a rank > 8,191: a producer keeps the plain layout for such a batch (SlabBuilder does)."""
    slab = make_slab(4, 100, seed=1, coverage=0.5)
    slab["rpr"][0, 0] = 9000
    with pytest.raises(ValueError):
        tag_ranks(slab)


def test_unknown_layout_bits_are_refused(bv):
    slab = make_slab(4, 100, seed=1, coverage=0.5)
    slab["layout"] = 0x2
    eng = bv.BaseTypeEngine(max_sites=4, min_af_value=bv.min_af(100), device=0)
    with pytest.raises(RuntimeError, match="layout"):
        eng.lrt(slab)
    eng.close()


@pytest.mark.parametrize("flags", [0, 0x8], ids=["joined_rows", "per_site_tallies"])
@pytest.mark.parametrize("n,width,groups", [(3001, 200, 0), (1000, 200, 2), (70000, 5000, 0), (10000, 1000, 0)])
def test_tagged_tiles_give_the_plain_layouts_records(bv, n, width, groups, flags):
    slab = make_slab(60, n, seed=900 + n % 13, coverage=0.1, n_groups=groups, ref_n_frac=0.02)
    maf = bv.min_af(n)
    res = []
    for s in (slab, tag_ranks(slab)):
        eng = bv.BaseTypeEngine(max_sites=60, min_af_value=maf, device=0, flags=flags)
        res.append(eng.lrt_tiles(s, width))
        eng.close()
    same(res[0], res[1])
    rows = run_engine(bv, tag_ranks(slab), maf)
    if flags == 0:
        same(rows, res[1])  # joined rows: the records of the row submit, bit for bit
    assert res[0].n_variant >= 2


def test_tagged_tile_job_with_fewer_samples_than_announced(bv):
    """Undelivered columns of a joined-rows job are uncovered cells: in the tagged layout their RANK WORDS must say so too."""
    from basevar_amd import _capi
    S, n, missing, width = 48, 9000, 1000, 1500
    slab = make_slab(S, n, seed=91, coverage=0.2, n_groups=0)
    full = dict(slab)
    for k, fill in (("base_strand", 8), ("qual", 0), ("mapq", 0), ("rpr", 0)):
        full[k] = np.concatenate([slab[k][:, :n], np.full((S, missing), fill, dtype=slab[k].dtype)], axis=1)
    full["n_samples"] = n + missing
    full["pitch"] = n + missing
    maf = bv.min_af(n + missing)
    want = run_engine(bv, full, maf)
    tg = tag_ranks(slab)
    eng = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0)
    assert eng._lib.bv_engine_tiles_begin(eng._h, S, n + missing, 0, 1) == 0, eng._err()
    keep = []
    P = (width + 15) // 16 * 16
    for lo in range(0, n, width):
        planes = []
        for k, dt, fill in (("base_strand", np.uint8, 8), ("qual", np.uint8, 0), ("mapq", np.uint8, 0), ("rpr", np.uint16, 0)):
            a = np.full((S, P), fill, dtype=dt)
            a[:, :width] = tg[k][:, lo:lo + width]
            planes.append(a)
        keep.append(planes)
        t = _capi.Slab(S, width, P, planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, planes[3].ctypes.data, None, None, 0,
                       _capi.BV_MEM_HOST, _capi.BV_SLAB_RPR_TAGGED)
        assert eng._lib.bv_engine_tiles_add(eng._h, C.byref(t), None) == 0, eng._err()
    # a tile of the other layout does not join this job
    t = _capi.Slab(S, width, P, planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, planes[3].ctypes.data, None, None, 0, _capi.BV_MEM_HOST, 0)
    assert eng._lib.bv_engine_tiles_add(eng._h, C.byref(t), None) == _capi.BV_ERR_INVALID_ARG
    ref = np.ascontiguousarray(slab["ref_base"], dtype=np.uint8)
    out = np.zeros(S, dtype=bv.SITE_DTYPE)
    assert eng._lib.bv_engine_tiles_finish(eng._h, ref.ctypes.data, out.ctypes.data, None, _capi.BV_MEM_HOST, None) == 0, eng._err()
    eng.wait()
    assert out.tobytes() == want.sites.tobytes()
    eng.close()


@pytest.mark.parametrize("n,groups", [(60000, 0), (6000, 0), (6000, 3), (1500, 0)], ids=["long_rows", "short_rows", "short_rows_groups", "rows_of_1500"])
def test_tagged_chained_submit(bv, n, groups):
    """bv_engine_submit_many on device slabs of the tagged layout = the plain layout's separate submits, byte for byte; a queue
    that mixes the layouts is submitted slab by slab and still gives them."""
    import torch
    sizes = [96, 17, 200, 64, 1, 130, 48]
    slabs = [make_slab(s, n, seed=300 + k, coverage=0.05 + 0.02 * (k % 3), class_af=[(0.0, 0.0), (0.3, 0.0), (0.2, 0.1)]) for k, s in enumerate(sizes)]
    slabs[5]["rpr"][3, np.nonzero(slabs[5]["base_strand"][3] < 8)[0][:3]] = 700  # a long read: the window sweeps
    maf = bv.min_af(n)
    dev = torch.device("cuda", 0)
    rec, grec = bv.SITE_DTYPE.itemsize, bv.GROUP_DTYPE.itemsize
    gid_t = None
    if groups:
        g = np.random.default_rng(n + groups).integers(0, groups + 1, size=slabs[0]["pitch"]).astype(np.uint8)
        g[g == groups] = 255
        gid_t = torch.from_numpy(g).to(dev)

    def run(layouts, chained):
        eng = bv.BaseTypeEngine(max_sites=sum(sizes), min_af_value=maf, device=0)
        keep, segs, outs, gouts = [], [], [], []
        for sl, lay in zip(slabs, layouts):
            s = tag_ranks(sl) if lay else sl
            t = [torch.from_numpy(np.ascontiguousarray(s[k])).to(dev) for k in ("base_strand", "qual", "ref_base", "mapq")]
            t.append(torch.from_numpy(np.ascontiguousarray(s["rpr"]).view(np.int16)).to(dev))
            out = torch.zeros(sl["n_sites"] * rec, dtype=torch.uint8, device=dev)
            gout = torch.zeros(max(1, sl["n_sites"] * groups * grec), dtype=torch.uint8, device=dev)
            keep.append(t); outs.append(out); gouts.append(gout)
            segs.append((sl["n_sites"], t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr(), t[3].data_ptr(), t[4].data_ptr()))
        torch.cuda.synchronize()
        if chained and len(set(layouts)) == 1:
            eng.submit_many_ptrs(n, slabs[0]["pitch"], segs, group_id=gid_t.data_ptr() if groups else 0, n_groups=groups,
                                 gouts=[g_.data_ptr() for g_ in gouts], layout=layouts[0])
        elif chained:
            from basevar_amd import _capi
            arr = (_capi.Slab * len(segs))()
            op = (C.c_void_p * len(segs))()
            gp = (C.c_void_p * len(segs))()
            for k, (ns, bs, q, ref, out, mq, rp) in enumerate(segs):
                arr[k] = _capi.Slab(ns, n, slabs[0]["pitch"], bs, q, mq, rp, ref, gid_t.data_ptr() if groups else None, groups, _capi.BV_MEM_DEVICE, layouts[k])
                op[k] = out
                gp[k] = gouts[k].data_ptr()
            assert eng._lib.bv_engine_submit_many_g(eng._h, len(segs), arr, op, gp if groups else None, None) == 0, eng._err()
        else:
            for k, sl in enumerate(slabs):
                eng.submit_ptrs(sl["n_sites"], n, sl["pitch"], segs[k][1], segs[k][2], segs[k][3], segs[k][4], segs[k][5], segs[k][6],
                                group_id=gid_t.data_ptr() if groups else 0, n_groups=groups, gout=gouts[k].data_ptr() if groups else 0, layout=layouts[k])
        eng.wait()
        r = [o.cpu().numpy().tobytes() for o in outs], [g_.cpu().numpy().tobytes() for g_ in gouts]
        eng.close()
        return r

    want = run([0] * len(sizes), chained=False)
    assert run([1] * len(sizes), chained=True) == want
    assert run([1] * len(sizes), chained=False) == want
    assert run([k & 1 for k in range(len(sizes))], chained=True) == want
    nv = sum(int(((np.frombuffer(b, dtype=bv.SITE_DTYPE)["status"] & 2) != 0).sum()) for b in want[0])
    assert nv > 50


def test_tagged_synthetic_generator(bv):
    """bv_synth_fill with layout = BV_SLAB_RPR_TAGGED writes exactly BV_RPR_TAGGED(call, rank) of the planes it writes without
    it, and the engine's records of the two slabs are byte-identical (both row lengths the bench uses)."""
    import torch
    from basevar_amd import _capi, synth_fill
    dev = torch.device("cuda", 0)
    for n, S in ((100000, 256), (10000, 2048)):
        P = (n + 15) // 16 * 16
        planes = {}
        for lay in (0, 1):
            bs = torch.empty((S, P), dtype=torch.uint8, device=dev); q = torch.empty_like(bs); mq = torch.empty_like(bs)
            rp = torch.empty((S, P), dtype=torch.int16, device=dev); ref = torch.empty(S, dtype=torch.uint8, device=dev)
            synth_fill(0, S, n, P, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), mq.data_ptr(), rp.data_ptr(), seed=5, site_offset=40, layout=lay)
            torch.cuda.synchronize()
            planes[lay] = (bs, q, mq, rp, ref)
        b0 = planes[0][0].cpu().numpy().astype(np.uint16)
        r0 = planes[0][3].cpu().numpy().view(np.uint16)
        r1 = planes[1][3].cpu().numpy().view(np.uint16)
        assert np.array_equal(planes[0][0].cpu().numpy(), planes[1][0].cpu().numpy())
        assert np.array_equal(r1, (r0 & 0x1FFF) | ((b0 & 3) << 13) | (((b0 >> 3) & 1) << 15))
        assert int(r0.max()) <= 100
        maf = bv.min_af(n)
        recs = []
        for lay in (0, 1):
            bs, q, mq, rp, ref = planes[lay]
            eng = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0)
            out = torch.zeros(S * bv.SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            eng.submit_ptrs(S, n, P, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), out.data_ptr(), mq.data_ptr(), rp.data_ptr(), layout=lay)
            eng.wait()
            recs.append(out.cpu().numpy().tobytes())
            eng.close()
        assert recs[0] == recs[1]
        assert int(((np.frombuffer(recs[0], dtype=bv.SITE_DTYPE)["status"] & 2) != 0).sum()) > S // 10


@pytest.mark.parametrize("n,sites,cov,flags", [(70000, 48, 1.0, 0), (70003, 48, 0.3, 0), (12000, 300, 1.0, 0), (12007, 300, 0.4, 0), (12000, 300, 1.0, 0x9000),
                                               (3000, 300, 1.0, 0), (12000, 200, 1.0, 1 << 16)],
                         ids=["long_rows_full", "long_rows_0.3", "fused_full", "fused_0.4", "three_launches_full", "dma_rows_full", "fused_one_workgroup"])
def test_deep_rows_count_the_dominant_mapq(bv, restatement, n, sites, cov, flags):
    """Deep rows (an eighth of the cells are REF / ALT reads) take bv_lds_add16_dom in the mapq tally of the rank sums: the
    lanes that hold the chunk's dominant value are counted and added once.  Records: those of the plain adds
    (BV_FLAG_NO_DOM), byte for byte, in both rank layouts -- and the oracle's."""
    slab = make_slab(sites, n, seed=4242 + n, coverage=cov, class_af=[(0.0, 0.0), (0.05, 0.0), (0.4, 0.0), (0.2, 0.1)], ref_n_frac=0.02)
    if cov == 1.0:
        slab["mapq"][5, :] = 60   # one value only; and a row whose first cells are NOT the dominant value
        slab["mapq"][6, :16] = np.arange(1, 17)
    maf = bv.min_af(n)
    plain, tagged = both(bv, slab, maf, flags)
    same(plain, tagged)
    no_dom, _ = both(bv, slab, maf, flags | 0x1000000)
    same(plain, no_dom)
    exp, gexp, margins = oracle_run(restatement, slab, maf)
    check(tagged, exp, gexp, margins)
    assert plain.n_variant >= sites // 3


def test_dense_long_rows_take_the_bank_swizzle(bv, restatement):
    """Long rows behind a dense row (a quarter of the cells covered) are tallied with the bank swizzle (bv_tally_chunk<.., SWZ>)
    and put back in order before the solver reads them: 3,000 fully covered rows of 60,000 samples (three per workgroup, so the
    second and third of each take it), generated on the device.  Records: those of BV_FLAG_NO_DOM (plain adds, no swizzle), byte
    for byte; a spread of sites against the oracle."""
    import torch
    from basevar_amd import synth_fill
    dev = torch.device("cuda", 0)
    S, n = 3000, 60000
    P = (n + 15) // 16 * 16
    bs = torch.empty((S, P), dtype=torch.uint8, device=dev); q = torch.empty_like(bs); mq = torch.empty_like(bs)
    rp = torch.empty((S, P), dtype=torch.int16, device=dev); ref = torch.empty(S, dtype=torch.uint8, device=dev)
    synth_fill(0, S, n, P, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), mq.data_ptr(), rp.data_ptr(), seed=77, site_offset=3, coverage=1.0, layout=1)
    torch.cuda.synchronize()
    maf = bv.min_af(n)
    recs = []
    for flags in (0, 0x1000000):
        eng = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0, flags=flags)
        out = torch.zeros(S * bv.SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        eng.submit_ptrs(S, n, P, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), out.data_ptr(), mq.data_ptr(), rp.data_ptr(), layout=1)
        eng.wait()
        recs.append(out.cpu().numpy().view(bv.SITE_DTYPE).copy())
        eng.close()
    assert recs[0].tobytes() == recs[1].tobytes()
    idx = np.linspace(0, S - 1, num=40).astype(np.int64)
    ti = torch.from_numpy(idx).to(dev)
    slab = {"base_strand": bs[ti].cpu().numpy(), "qual": q[ti].cpu().numpy(), "mapq": mq[ti].cpu().numpy(),
            "rpr": rp[ti].cpu().numpy().view(np.uint16) & np.uint16(0x1FFF), "ref_base": ref[ti].cpu().numpy(), "n_samples": n, "pitch": P, "n_sites": 40}
    exp, gexp, margins = oracle_run(restatement, slab, maf)

    class Got:
        sites = recs[0][idx]; groups = None; n_variant = int(((recs[0][idx]["status"] & 2) != 0).sum())
    check(Got, exp, gexp, margins)
    assert (recs[0]["total_depth"] > n * 0.9).all()


@pytest.mark.parametrize("flags", [0, 0x8], ids=["joined_rows", "per_site_tallies"])
@pytest.mark.parametrize("n,width,groups,tagged", [(3001, 200, 0, False), (1000, 200, 2, True), (70000, 5000, 0, True), (10000, 1000, 3, False), (640, 64, 0, False)])
def test_packed_host_tiles_give_the_dense_tiles_records(bv, n, width, groups, tagged, flags):
    """bv_engine_tiles_add_sparse: a tile as its covered cells only (7 bytes per covered cell instead of 5 per cell over the host
    link) -- the records of the job must be those of the dense tiles, byte for byte, in both realisations, with and without the
    tagged rank layout, pop-groups included, and in a job that mixes dense and packed tiles (every third tile dense)."""
    slab = make_slab(60, n, seed=1900 + n % 13, coverage=0.1, n_groups=groups, ref_n_frac=0.02)
    slab["rpr"][7, np.nonzero(slab["base_strand"][7] < 8)[0][:5]] = 700   # long reads: window sweeps / the overflow pool
    if tagged:
        slab = tag_ranks(slab)
    maf = bv.min_af(n)
    res = []
    for packed in (False, True, 3):
        eng = bv.BaseTypeEngine(max_sites=60, min_af_value=maf, device=0, flags=flags)
        res.append(eng.lrt_tiles(slab, width, packed=packed))
        eng.close()
    same(res[0], res[1])
    same(res[0], res[2])
    if flags == 0:
        same(run_engine(bv, slab, maf), res[1])
    assert res[0].n_variant >= 2


def test_packed_tile_api_errors_and_an_empty_tile(bv):
    from basevar_amd import _capi
    S, n, w = 32, 600, 200
    slab = make_slab(S, n, seed=5, coverage=0.2)
    slab["base_strand"][:, 200:400] = 8   # the second tile: nobody covered -> a packed tile without entries
    for k in ("qual", "mapq", "rpr"):
        slab[k][:, 200:400] = 0
    maf = bv.min_af(n)
    want = run_engine(bv, slab, maf)
    eng = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0)
    got = eng.lrt_tiles(slab, w, packed=True)
    same(want, got)
    # a tile of more than 65,536 samples has no 16-bit sample index; a job must be open
    rs = np.zeros(S + 1, dtype=np.uint32)
    t = _capi.SparseTile(S, 70000, 0, 0, rs.ctypes.data, None, None, None, None, None, None, _capi.BV_MEM_HOST, 0)
    assert eng._lib.bv_engine_tiles_add_sparse(eng._h, C.byref(t), None) == _capi.BV_ERR_INVALID_ARG
    assert eng._lib.bv_engine_tiles_begin(eng._h, S, 100000, 0, 1) == 0
    assert eng._lib.bv_engine_tiles_add_sparse(eng._h, C.byref(t), None) == _capi.BV_ERR_INVALID_ARG
    eng.close()


@pytest.mark.parametrize("n,G,cov,ranks", [(10000, 32, 0.08, True), (10000, 64, 0.08, True), (6000, 12, 0.05, False), (20000, 40, 0.1, True), (3000, 9, 0.3, True)],
                         ids=["g32", "g64_two_rounds", "g12_no_rank_planes", "g40_mixed_with_big_groups", "g9_deep_groups"])
def test_small_pop_groups_through_the_small_solvers(bv, restatement, n, G, cov, ranks):
    """8 and more pop-groups on short rows: the items of small groups go through bv_p2g_solve_small_kernel -- 4 or 8 lanes per
    item, sixteen or eight items per wave, the kind chosen per site by the tally kernel (BV_P2G_L4 / BV_P2G_L8) --, bigger groups
    keep the 16-lane solver; the fused kernel streams the rank-sum rows whatever the number of groups.  Against the oracle."""
    slab = make_slab(300, n, seed=77 + G, coverage=cov, n_groups=G, ref_n_frac=0.02, site_offset=11)
    if not ranks:
        slab.pop("mapq"); slab.pop("rpr")
    if G == 40:
        slab["group_id"][: n // 2] = 0      # one group holds half the cohort: its items are big beside 39 small ones
    maf = bv.min_af(n)
    got = run_engine(bv, slab, maf)
    exp, gexp, margins = oracle_run(restatement, slab, maf)
    check(got, exp, gexp, margins, check_ranks=ranks)
    assert got.n_variant >= 20


def test_dense_long_rows_of_binned_qualities_and_one_value(bv, restatement):
    """Dense long rows whose cells sit on a handful of (strand, base, phred) words -- a sequencer that bins its qualities (2, 12,
    23, 37), and the degenerate row of ONE call and ONE phred in every sample: the worst case of the LDS tallies (one hot word per
    strand).  Slow, but exact: records those of BV_FLAG_NO_DOM (plain adds, no swizzle), byte for byte, and the oracle's on a spread
    of sites."""
    import torch
    dev = torch.device("cuda", 0)
    S, n = 2600, 50000
    P = (n + 15) // 16 * 16
    g = torch.Generator(device="cpu").manual_seed(99)
    ref = torch.randint(0, 4, (S,), generator=g, dtype=torch.uint8)
    strand = torch.randint(0, 2, (S, P), generator=g, dtype=torch.uint8)
    alt = torch.rand((S, P), generator=g) < 0.03
    base = torch.where(alt, (ref[:, None] + 1) % 4, ref[:, None].expand(S, P)).to(torch.uint8)
    bs = (base | (strand << 2)).to(torch.uint8)
    qb = torch.tensor([2, 12, 23, 37], dtype=torch.uint8)[torch.randint(0, 4, (S, P), generator=g)]
    qb = torch.where(torch.rand((S, P), generator=g) < 0.7, torch.full_like(qb, 37), qb)
    # rows 0-99: every sample the same call and phred (forward reference base, phred 37)
    bs[:100] = ref[:100, None]
    qb[:100] = 37
    mq = torch.full((S, P), 60, dtype=torch.uint8)
    rp = torch.randint(1, 101, (S, P), generator=g, dtype=torch.int16)
    bs_d, q_d, mq_d, rp_d, ref_d = (t.to(dev).contiguous() for t in (bs, qb, mq, rp, ref))
    maf = bv.min_af(n)
    recs = []
    for flags in (0, 0x1000000):
        eng = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0, flags=flags)
        out = torch.zeros(S * bv.SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        eng.submit_ptrs(S, n, P, bs_d.data_ptr(), q_d.data_ptr(), ref_d.data_ptr(), out.data_ptr(), mq_d.data_ptr(), rp_d.data_ptr())
        eng.wait()
        recs.append(out.cpu().numpy().view(bv.SITE_DTYPE).copy())
        eng.close()
    assert recs[0].tobytes() == recs[1].tobytes()
    idx = np.concatenate([np.arange(0, 6), np.linspace(100, S - 1, num=26).astype(np.int64)])
    slab = {"base_strand": bs[idx].numpy(), "qual": qb[idx].numpy(), "mapq": mq[idx].numpy(), "rpr": rp[idx].numpy().view(np.uint16),
            "ref_base": ref[idx].numpy(), "n_samples": n, "pitch": P, "n_sites": len(idx)}
    exp, gexp, margins = oracle_run(restatement, slab, maf)

    class Got:
        sites = recs[0][idx]; groups = None; n_variant = int(((recs[0][idx]["status"] & 2) != 0).sum())
    check(Got, exp, gexp, margins)
    assert (recs[0]["total_depth"] == n).all()


@pytest.mark.parametrize("n,sites,groups,flags", [
    (10000, 900, 0, 0), (10007, 400, 0, 0), (4097, 300, 0, 0), (5120, 300, 0, 0), (6145, 200, 0, 0), (16384, 150, 0, 0), (16385, 150, 0, 0),
    (49151, 64, 0, 0), (12000, 400, 9, 0), (12000, 300, 0, 1 << 16), (20011, 300, 0, 2 << 16), (10000, 6000, 0, 4 << 16)],
    ids=["configs1", "ragged", "shortest", "5120", "6145", "16384", "16385", "longest", "groups9", "one_workgroup", "two_workgroups", "four_workgroups_6000"])
def test_fused_pass2_rows_in_registers_or_through_the_ring(bv, restatement, n, sites, groups, flags):
    """Behind their pass-1 rows the waves of the fused short-row kernel -- the streaming waves and, out of jobs, the solver waves --
    tally the variant sites' rank-sum rows of a TAGGED slab from registers (bv_f_p2_rows, round 6: plain loads, no ring);
    BV_FLAG_P2_TAIL_DMA keeps those rows in the LDS-DMA ring, where the plain layout's rows always are.  Records: byte-identical
    between the two paths and the two rank layouts, at every block boundary of a row (1,024 cells per block, partial last blocks
    and chunks), with pop-groups, with every queue overflowing (one workgroup), with thousands of sites per workgroup -- and the
    oracle's.  Long reads (ranks >= 256: the row is re-done by the window sweeps) too."""
    slab = make_slab(sites, n, seed=977 + n, coverage=0.08 if n > 6000 else 0.3, class_af=[(0.0, 0.0), (0.05, 0.0), (0.4, 0.0), (0.2, 0.1)],
                     ref_n_frac=0.02, n_groups=groups)
    slab["rpr"][3::7, :] = np.minimum(slab["rpr"][3::7, :].astype(np.int64) * 9, 8000).astype(slab["rpr"].dtype)   # every seventh row: long reads
    maf = bv.min_af(n)
    plain, tagged = both(bv, slab, maf, flags)
    same(plain, tagged)
    ring_plain, ring_tagged = both(bv, slab, maf, flags | 0x2000000)
    same(plain, ring_plain)
    same(tagged, ring_tagged)
    exp, gexp, margins = oracle_run(restatement, slab, maf)
    check(tagged, exp, gexp, margins)
    assert plain.n_variant >= sites // 4
