"""Host-side mirror of the reference's per-site interface, batched.

The reference calls, once per genomic site (src/basetype_caller.cpp:742-743, 1113-1164):

    BaseType bt(&batchinfo, min_af);  bt.lrt();
    bt.get_alt_bases(); bt.get_lrt_af(b); bt.get_var_qual(); bt.get_total_depth(); bt.get_base_depth(b)
    strand_bias(ref, alts, bases, strands);  ref_vs_alt_ranksumtest(ref, alts, bases, values)

``BaseTypeEngine.lrt(slab)`` does all of that for every site (row) of a slab in one submit
through the C ABI (include/basevar_amd.h) and returns ``BaseTypeBatch``, whose getters carry
the reference's names and take the site index as first argument.  Errors the reference
raises as std::runtime_error surface as RuntimeError with the same message.

torch is used only for device memory and streams; it never appears in the ABI.
"""
import ctypes as C

import numpy as np

from . import _capi

BASES = "ACGT"  # src/basetype.h:19


def min_af(n_samples, user_min_af=0.01):  # noqa: D401
    """(double)std::min(float(100)/n, min_af): src/basetype_caller.cpp:122."""
    return _capi.load().bv_min_af(int(n_samples), float(user_min_af))


class BaseTypeBatch:
    """Per-site records of one submit (numpy structured arrays on the host)."""

    def __init__(self, sites, groups, n_variant, pass1_ms, pass2_ms):
        self.sites = sites
        self.groups = groups
        self.n_variant = n_variant
        self.pass1_ms = pass1_ms
        self.pass2_ms = pass2_ms

    # --- BaseType getters, src/basetype.h:121-151
    def get_alt_bases(self, i):
        r = self.sites[i]
        return [BASES[b] for b in r["alt"][:r["n_alt"]]]

    def get_lrt_af(self, i, b):
        r = self.sites[i]
        alts = [BASES[x] for x in r["alt"][:r["n_alt"]]]
        if b not in alts:  # std::map::at -> out_of_range -> runtime_error, basetype.h:141-149
            raise RuntimeError("[ERROR] out_of_range:: map::at '%s' not found." % b)
        return float(r["af"][alts.index(b)])

    def get_var_qual(self, i):
        return float(self.sites[i]["qual"])

    def get_total_depth(self, i):
        return int(self.sites[i]["total_depth"])

    def get_base_depth(self, i, b):
        if b not in BASES:
            raise RuntimeError("[ERROR] out_of_range:: map::at '%s' not found." % b)
        return float(self.sites[i]["depth"][BASES.index(b)])

    # --- StrandBiasInfo of the two strand_bias() calls, src/basetype.h:57-62
    def strand_bias(self, i, flavour="vcf"):
        r = self.sites[i]
        k = "var" if flavour == "vcf" else "cvg"
        sb = r[k + "_sb"]
        return {"ref_fwd": int(sb[0]), "ref_rev": int(sb[1]), "alt_fwd": int(sb[2]), "alt_rev": int(sb[3]),
                "fs": float(r[k + "_fs"]), "sor": float(r[k + "_sor"])}

    # --- the three ref_vs_alt_ranksumtest() values; the reference truncates them to int
    def rank_sums(self, i):
        r = self.sites[i]
        return float(r["mq_ranksum"]), float(r["rpr_ranksum"]), float(r["bq_ranksum"])


class BaseTypeEngine:
    """One engine per GPU / host thread (mirrors one BaseType per ThreadPool worker)."""

    def __init__(self, max_sites, min_af_value, device=0, flags=0, max_samples=0):
        self._lib = _capi.load()
        cfg = _capi.EngineConfig(int(device), int(max_sites), int(max_samples), int(flags), float(min_af_value))
        h = C.c_void_p()
        rc = self._lib.bv_engine_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError("bv_engine_create failed (%d): %s" % (rc, self._lib.bv_last_error(None).decode()))
        self._h = h
        self.device = int(device)
        self.max_sites = int(max_sites)
        self.min_af = float(min_af_value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bv_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return self._lib.bv_last_error(self._h).decode()

    # ---- raw pointer interface (device or host pointers as ints)
    def submit_ptrs(self, n_sites, n_samples, pitch, base_strand, qual, ref_base, out, mapq=0, rpr=0, group_id=0,
                    n_groups=0, gout=0, mem_kind=_capi.BV_MEM_DEVICE, stream=0, layout=0):
        """`layout`: bv_slab.layout (BV_SLAB_RPR_TAGGED: the rpr plane carries the cells' calls, include/basevar_amd.h)."""
        slab = _capi.Slab(int(n_sites), int(n_samples), int(pitch), base_strand or None, qual or None, mapq or None,
                          rpr or None, ref_base or None, group_id or None, int(n_groups), int(mem_kind), int(layout))
        rc = self._lib.bv_engine_submit(self._h, C.byref(slab), out or None, gout or None, stream or None)
        if rc != 0:
            raise RuntimeError("bv_engine_submit failed (%d): %s" % (rc, self._err()))

    def submit_many_ptrs(self, n_samples, pitch, slabs, stream=0, group_id=0, n_groups=0, gouts=None, layout=0):
        """Several device-resident slabs as ONE launch per pass (bv_engine_submit_many / _g).  `slabs`: a sequence of
        (n_sites, base_strand, qual, ref_base, out, mapq, rpr) with device pointers as ints (mapq = rpr = 0: no rank sums);
        pop-groups: one `group_id` array for the whole queue, `gouts[k]` = slab k's group records."""
        n = len(slabs)
        arr = (_capi.Slab * n)()
        outs = (C.c_void_p * n)()
        gp = (C.c_void_p * n)() if n_groups else None
        for k, (n_sites, bs, q, ref, out, mq, rp) in enumerate(slabs):
            arr[k] = _capi.Slab(int(n_sites), int(n_samples), int(pitch), bs or None, q or None, mq or None, rp or None, ref or None,
                                group_id or None, int(n_groups), _capi.BV_MEM_DEVICE, int(layout))
            outs[k] = out
            if n_groups:
                gp[k] = gouts[k]
        rc = self._lib.bv_engine_submit_many_g(self._h, n, arr, outs, gp, stream or None)
        if rc != 0:
            raise RuntimeError("bv_engine_submit_many failed (%d): %s" % (rc, self._err()))

    def join(self, stream=0):
        """Make `stream` wait for every submit issued so far (needed with BV_FLAG_LANES: bv_engine_join)."""
        if self._lib.bv_engine_join(self._h, C.c_void_p(stream) if stream else None) != 0:
            raise RuntimeError("bv_engine_join: " + self._err())

    def stream_handle(self):
        """hipStream_t of the engine's own stream as an int (e.g. for torch.cuda.ExternalStream).  The stream dies with the
        engine: free torch tensors that were used on it -- pinned host blocks above all, whose allocator records an event on
        every stream a block has seen when the block is freed -- before close()."""
        return int(self._lib.bv_engine_stream(self._h) or 0)

    def wait(self):
        rc = self._lib.bv_engine_wait(self._h)
        if rc == _capi.BV_ERR_SITE:
            raise RuntimeError(self._err())  # the reference's runtime_error text, basetype.cpp:114
        if rc != 0:
            raise RuntimeError("bv_engine_wait failed (%d): %s" % (rc, self._err()))

    def kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        rc = self._lib.bv_engine_kernel_ms(self._h, C.byref(a), C.byref(b))
        if rc != 0:
            raise RuntimeError("bv_engine_kernel_ms failed (%d): %s" % (rc, self._err()))
        return a.value, b.value

    def timing_reset(self):
        if self._lib.bv_engine_timing_reset(self._h) != 0:
            raise RuntimeError("bv_engine_timing_reset: " + self._err())

    def timing_get(self):
        """(total pass-1 ms, total pass-2 ms, number of submits) since timing_reset()."""
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        if self._lib.bv_engine_timing_get(self._h, C.byref(a), C.byref(b), C.byref(n)) != 0:
            raise RuntimeError("bv_engine_timing_get: " + self._err())
        return a.value, b.value, n.value

    def timing_get_ex(self):
        """(streaming-kernel ms, pass-1 ms, pass-2 ms, submits) since timing_reset(); on short rows pass 1 is the
        streaming kernel plus the solve kernel, on long rows it is one kernel and the first two figures coincide."""
        s, a, b, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        if self._lib.bv_engine_timing_get_ex(self._h, C.byref(s), C.byref(a), C.byref(b), C.byref(n)) != 0:
            raise RuntimeError("bv_engine_timing_get_ex: " + self._err())
        return s.value, a.value, b.value, n.value

    @property
    def host_log_exact(self):
        """True when shallow sites are replayed with the host libm's own log() (verified bit-exact at creation)."""
        return bool(self._lib.bv_engine_host_log_exact(self._h))

    def host_log_eval(self, x):
        """The device restatement of the host's log() at the float64 array x (diagnostic)."""
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        if self._lib.bv_engine_host_log_eval(self._h, x.ctypes.data, y.ctypes.data, x.size) != 0:
            raise RuntimeError("bv_engine_host_log_eval: " + self._err())
        return y

    def last_launch_form(self):
        """BV_FORM_* bits of the last launch's pass 1 (include/basevar_amd_diag.h)."""
        f = C.c_uint32()
        if self._lib.bv_engine_last_launch_form(self._h, C.byref(f)) != 0:
            raise RuntimeError("bv_engine_last_launch_form: " + self._err())
        return f.value

    def last_variant_count(self):
        n = C.c_uint32()
        self._lib.bv_engine_last_variant_count(self._h, C.byref(n))
        return n.value

    # ---- sample-axis tile mode (numpy tiles in host memory)
    def lrt_tiles(self, slab, tile_width, max_rank=0, packed=False):
        """`packed`: every tile goes as its covered cells only (bv_engine_tiles_add_sparse: 7 bytes per covered cell, one packed
        host allocation per tile); a number k > 1: every k-th tile dense, the others packed (a job may mix the two).
        Same result as lrt(slab), but the slab is fed as column tiles of `tile_width` samples
        (the reference's `-B/--batch-count` batchfiles) that the engine accumulates in HBM.  Every tile is one packed
        host allocation (bv_tile_packed_layout), so it crosses the link as one copy.  `max_rank`: an upper bound on the
        read-position ranks, for the per-site-tally realisation (see bv_engine_tiles_begin)."""
        bs = np.ascontiguousarray(slab["base_strand"], dtype=np.uint8)
        S = bs.shape[0]
        N = int(slab.get("n_samples", bs.shape[1]))
        q = np.asarray(slab["qual"], dtype=np.uint8)
        mq, rp = slab.get("mapq"), slab.get("rpr")
        ranks = mq is not None and rp is not None
        gid = slab.get("group_id")
        ng = int(slab.get("n_groups", 0)) if gid is not None else 0
        ref = np.ascontiguousarray(slab["ref_base"], dtype=np.uint8)
        rc = self._lib.bv_engine_tiles_begin(self._h, S, N, ng, (max(2, int(max_rank)) if max_rank else 1) if ranks else 0)
        if rc != 0:
            raise RuntimeError("bv_engine_tiles_begin failed (%d): %s" % (rc, self._err()))
        keep = []  # the copies are asynchronous: every tile stays alive until the final wait
        lay = int(slab.get("layout", 0))
        for k_tile, lo in enumerate(range(0, N, tile_width)):
            w = min(tile_width, N - lo)
            if packed and not (packed > 1 and k_tile % int(packed) == 0):
                # the tile's covered cells, site after site (np.nonzero walks row-major: sites in order, samples ascending)
                cb = bs[:, lo:lo + w]
                rows, cols = np.nonzero(cb != 8)
                E = int(rows.size)
                offs = (C.c_uint64 * 7)()
                total = C.c_uint64()
                rc = self._lib.bv_sparse_tile_packed_layout(S, E, w, 1 if ranks else 0, 1 if ng else 0, offs, C.byref(total))
                if rc != 0:
                    raise RuntimeError("bv_sparse_tile_packed_layout failed (%d)" % rc)
                buf = np.zeros(total.value + 256, dtype=np.uint8)
                pad = (-buf.ctypes.data) % 256

                def arr(kk, dt, n):
                    return buf[pad + offs[kk]: pad + offs[kk] + n * np.dtype(dt).itemsize].view(dt)
                rs = arr(0, np.uint32, S + 1); rs[0] = 0; rs[1:] = np.cumsum(np.bincount(rows, minlength=S))
                a_s = arr(1, np.uint16, E); a_s[:] = cols
                a_b = arr(2, np.uint8, E); a_b[:] = cb[rows, cols]
                a_q = arr(3, np.uint8, E); a_q[:] = q[:, lo:lo + w][rows, cols]
                a_m = a_r = a_g = None
                if ranks:
                    a_m = arr(4, np.uint8, E); a_m[:] = np.asarray(mq)[:, lo:lo + w][rows, cols]
                    a_r = arr(5, np.uint16, E); a_r[:] = np.asarray(rp)[:, lo:lo + w][rows, cols] & (0x1FFF if lay & 1 else 0xFFFF)  # plain ranks
                if ng:
                    a_g = arr(6, np.uint8, w); a_g[:] = np.asarray(gid, dtype=np.uint8)[lo:lo + w]
                keep.append(buf)
                p = lambda a: None if a is None else a.ctypes.data
                t = _capi.SparseTile(S, w, E, ng, p(rs), p(a_s), p(a_b), p(a_q), p(a_m), p(a_r), p(a_g), _capi.BV_MEM_HOST, lay)
                rc = self._lib.bv_engine_tiles_add_sparse(self._h, C.byref(t), None)
                if rc != 0:
                    raise RuntimeError("bv_engine_tiles_add_sparse failed (%d): %s" % (rc, self._err()))
                if len(keep) >= 64:
                    self.wait()
                    del keep[:-1]
                continue
            pitch, total = C.c_uint64(), C.c_uint64()
            offs = (C.c_uint64 * 5)()
            rc = self._lib.bv_tile_packed_layout(S, w, 1 if ranks else 0, 1 if ng else 0, C.byref(pitch), offs, C.byref(total))
            if rc != 0:
                raise RuntimeError("bv_tile_packed_layout failed (%d)" % rc)
            P = pitch.value
            buf = np.zeros(total.value + 256, dtype=np.uint8)
            base = buf.ctypes.data
            pad = (-base) % 256  # 256-byte aligned start inside the numpy allocation

            def plane(k, dt, rows, fill):
                n = rows * P * np.dtype(dt).itemsize
                v = buf[pad + offs[k]: pad + offs[k] + n].view(dt).reshape(rows, P)
                v[...] = fill
                return v
            tb = plane(0, np.uint8, S, 8); tb[:, :w] = bs[:, lo:lo + w]
            tq = plane(1, np.uint8, S, 0); tq[:, :w] = q[:, lo:lo + w]
            tm = tr = tg = None
            if ranks:
                tm = plane(2, np.uint8, S, 0); tm[:, :w] = np.asarray(mq)[:, lo:lo + w]
                tr = plane(3, np.uint16, S, 0); tr[:, :w] = np.asarray(rp)[:, lo:lo + w]
            if ng:
                tg = plane(4, np.uint8, 1, 0xFF); tg[0, :w] = np.asarray(gid, dtype=np.uint8)[lo:lo + w]
            keep.append(buf)
            p = lambda a: None if a is None else a.ctypes.data
            t = _capi.Slab(S, w, P, p(tb), p(tq), p(tm), p(tr), None, p(tg), ng, _capi.BV_MEM_HOST, int(slab.get("layout", 0)))
            rc = self._lib.bv_engine_tiles_add(self._h, C.byref(t), None)
            if rc != 0:
                raise RuntimeError("bv_engine_tiles_add failed (%d): %s" % (rc, self._err()))
            if len(keep) >= 64:  # bound the host memory held by in-flight tiles
                self.wait()
                del keep[:-1]
        out = np.zeros(S, dtype=_capi.SITE_DTYPE)
        gout = np.zeros((S, ng), dtype=_capi.GROUP_DTYPE) if ng else None
        rc = self._lib.bv_engine_tiles_finish(self._h, ref.ctypes.data, out.ctypes.data,
                                              gout.ctypes.data if ng else None, _capi.BV_MEM_HOST, None)
        if rc != 0:
            raise RuntimeError("bv_engine_tiles_finish failed (%d): %s" % (rc, self._err()))
        self.wait()
        return BaseTypeBatch(out, gout, self.last_variant_count(), 0.0, 0.0)

    def tiles_add_many(self, tiles, stream=0):
        """tiles: list of _capi.Slab (one open tile job, bv_engine_tiles_begin): device-resident tiles of a joined-rows job go
        to their columns in one launch per 256 tiles (bv_engine_tiles_add_many)."""
        arr = (_capi.Slab * len(tiles))(*tiles)
        rc = self._lib.bv_engine_tiles_add_many(self._h, len(tiles), arr, C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise RuntimeError("bv_engine_tiles_add_many failed (%d): %s" % (rc, self._err()))

    # ---- numpy slab (host memory; the engine stages it to HBM)
    def lrt(self, slab):
        """slab: dict of numpy planes as produced by basevar_amd.synth.make_slab()."""
        bs = np.ascontiguousarray(slab["base_strand"], dtype=np.uint8)
        S, pitch = bs.shape
        N = int(slab.get("n_samples", pitch))
        q = np.ascontiguousarray(slab["qual"], dtype=np.uint8)
        mq = slab.get("mapq")
        rp = slab.get("rpr")
        mq = None if mq is None else np.ascontiguousarray(mq, dtype=np.uint8)
        rp = None if rp is None else np.ascontiguousarray(rp, dtype=np.uint16)
        ref = np.ascontiguousarray(slab["ref_base"], dtype=np.uint8)
        gid = slab.get("group_id")
        ng = int(slab.get("n_groups", 0)) if gid is not None else 0
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.uint8)
        out = np.zeros(S, dtype=_capi.SITE_DTYPE)
        gout = np.zeros((S, ng), dtype=_capi.GROUP_DTYPE) if ng else None
        p = lambda a: 0 if a is None else a.ctypes.data
        self.submit_ptrs(S, N, pitch, p(bs), p(q), p(ref), p(out), p(mq), p(rp), p(gid), ng, p(gout),
                         mem_kind=_capi.BV_MEM_HOST, layout=int(slab.get("layout", 0)))
        self.wait()
        ms1, ms2 = self.kernel_ms()
        return BaseTypeBatch(out, gout, self.last_variant_count(), ms1, ms2)


def tile_packed_layout(n_sites, width, with_ranks=True, with_groups=False):
    """(pitch, [offsets of base_strand, qual, mapq, rpr, group_id], total bytes) of a packed host tile: one allocation
    that bv_engine_tiles_add sends over the link as one copy (include/basevar_amd.h)."""
    lib = _capi.load()
    pitch, total = C.c_uint64(), C.c_uint64()
    offs = (C.c_uint64 * 5)()
    rc = lib.bv_tile_packed_layout(n_sites, width, 1 if with_ranks else 0, 1 if with_groups else 0, C.byref(pitch), offs,
                                   C.byref(total))
    if rc != 0:
        raise RuntimeError("bv_tile_packed_layout failed (%d)" % rc)
    return pitch.value, [int(o) for o in offs], total.value


def synth_fill(device, n_sites, n_samples, pitch, base_strand, qual, ref_base, mapq=0, rpr=0, seed=0xBA5E7A7,
               site_offset=0, coverage=0.08, indel_frac=0.005, qual_mean=32.0, qual_sd=6.0, qual_min=2, qual_max=41,
               stream=0, layout=0):
    """Device-side synthetic pileup (bench helper): all pointers are device pointers (ints)."""
    lib = _capi.load()
    sp = _capi.SynthParams(int(seed), int(site_offset), float(coverage), float(indel_frac), float(qual_mean),
                           float(qual_sd), int(qual_min), int(qual_max), int(layout))
    rc = lib.bv_synth_fill(int(device), C.byref(sp), int(n_sites), int(n_samples), int(pitch), base_strand, qual,
                           mapq or None, rpr or None, ref_base, stream or None)
    if rc != 0:
        raise RuntimeError("bv_synth_fill failed (%d): %s" % (rc, lib.bv_last_error(None).decode()))
