"""ctypes declarations of the C ABI in include/basevar_amd.h.

The shared library is the product: it is built in-tree by ``__graft_entry__.build()``
(``make -C basevar_amd/csrc``) and lives at basevar_amd/lib/libbasevar_amd.so.  There is
no Python or CPU implementation behind this module -- if the library is missing, loading
fails loudly.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BASEVAR_AMD_LIB") or os.path.join(HERE, "lib", "libbasevar_amd.so")  # override: A/B builds

BV_MAX_ALT = 4
BV_MAX_GROUPS = 255
BV_NO_GROUP = 0xFF
BV_MEM_DEVICE, BV_MEM_HOST = 0, 1
BV_SLAB_RPR_TAGGED, BV_RPR_TAG_MAX_RANK = 0x1, 0x1FFF  # bv_slab.layout
BV_FLAG_LANES = 0x10000000
BV_FLAG_SPARSE_TIMING = 0x20000000  # include/basevar_amd.h
BV_FORM_SHORT_ROWS, BV_FORM_ONE_KERNEL, BV_FORM_PASS2_FUSED = 0x1, 0x2, 0x4  # include/basevar_amd_diag.h
# bv_engine_config.flags (include/basevar_amd.h)
BV_FLAG_TALLY_ONLY, BV_FLAG_SKIP_FISHER, BV_FLAG_SKIP_LRT, BV_FLAG_TILE_STATE, BV_FLAG_WAVE_SOLVER = 0x1, 0x2, 0x4, 0x8, 0x10
BV_OK, BV_ERR_INVALID_ARG, BV_ERR_NO_DEVICE, BV_ERR_HIP, BV_ERR_TOO_LARGE, BV_ERR_SITE = 0, -1, -2, -3, -4, -5

BV_SITE_COVERED, BV_SITE_VARIANT, BV_SITE_BAD_QUAL = 0x1, 0x2, 0x4
BV_SITE_ZERO_FREQ, BV_SITE_RANKSUM, BV_SITE_SOR_OVERFLOW, BV_SITE_RPR_RANGE = 0x8, 0x10, 0x20, 0x40

# cell encoding of the base_strand plane
BV_CELL_REV, BV_CELL_NOCALL, BV_CELL_N, BV_CELL_INS, BV_CELL_DEL = 0x04, 0x08, 0x08, 0x09, 0x0A

SITE_DTYPE = np.dtype([
    ("depth", "<u4", 4), ("total_depth", "<u4"), ("status", "<u4"),
    ("cvg_sb", "<u4", 4), ("cvg_fs", "<f8"), ("cvg_sor", "<f8"),
    ("n_alt", "u1"), ("alt", "u1", 4), ("n_em", "u1"), ("em_iters", "<u2"),
    ("af", "<f8", 4), ("caf", "<f8", 4), ("qual", "<f8"), ("chi2", "<f8"), ("qd", "<f8"),
    ("var_sb", "<u4", 4), ("var_fs", "<f8"), ("var_sor", "<f8"),
    ("mq_ranksum", "<f8"), ("rpr_ranksum", "<f8"), ("bq_ranksum", "<f8"),
])
GROUP_DTYPE = np.dtype([("n_alt", "u1"), ("alt", "u1", 4), ("reserved", "u1", 3), ("total_depth", "<u4"),
                        ("reserved2", "<u4"), ("af", "<f8", 4)])
assert SITE_DTYPE.itemsize == 208 and GROUP_DTYPE.itemsize == 48


class Slab(C.Structure):
    _fields_ = [("n_sites", C.c_uint32), ("n_samples", C.c_uint32), ("pitch", C.c_uint64),
                ("base_strand", C.c_void_p), ("qual", C.c_void_p), ("mapq", C.c_void_p), ("rpr", C.c_void_p),
                ("ref_base", C.c_void_p), ("group_id", C.c_void_p), ("n_groups", C.c_uint32),
                ("mem_kind", C.c_uint32), ("layout", C.c_uint32), ("reserved_", C.c_uint32)]


class SparseTile(C.Structure):
    _fields_ = [("n_sites", C.c_uint32), ("n_samples", C.c_uint32), ("n_entries", C.c_uint32), ("n_groups", C.c_uint32),
                ("row_start", C.c_void_p), ("sample", C.c_void_p), ("base_strand", C.c_void_p), ("qual", C.c_void_p), ("mapq", C.c_void_p),
                ("rpr", C.c_void_p), ("group_id", C.c_void_p), ("mem_kind", C.c_uint32), ("layout", C.c_uint32)]


class EngineConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_sites", C.c_uint32), ("max_samples", C.c_uint32),
                ("flags", C.c_uint32), ("min_af", C.c_double)]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("site_offset", C.c_uint64), ("coverage", C.c_float),
                ("indel_frac", C.c_float), ("qual_mean", C.c_float), ("qual_sd", C.c_float),
                ("qual_min", C.c_uint32), ("qual_max", C.c_uint32), ("layout", C.c_uint32), ("reserved_", C.c_uint32)]


# every symbol include/basevar_amd.h declares
EXPORTS = ["bv_version", "bv_min_af", "bv_engine_create", "bv_engine_destroy", "bv_engine_submit", "bv_engine_submit_many", "bv_engine_submit_many_g", "bv_engine_wait", "bv_engine_join",
           "bv_engine_tiles_begin", "bv_engine_tiles_add", "bv_engine_tiles_add_many", "bv_engine_tiles_add_sparse", "bv_sparse_tile_packed_layout", "bv_engine_tiles_finish", "bv_tile_packed_layout", "bv_engine_stream",
           "bv_engine_kernel_ms", "bv_engine_timing_reset", "bv_engine_timing_get", "bv_engine_timing_get_ex",
           "bv_host_log_probe", "bv_host_log_eval", "bv_engine_host_log_exact", "bv_engine_host_log_eval",
           "bv_engine_last_variant_count", "bv_last_error", "bv_synth_fill", "bv_device_numa_node", "bv_bind_thread_to_device_node", "bv_engine_last_launch_form"]

_lib = None


def load():
    """Load the C-ABI library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "basevar_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C basevar_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If this
    # library were loaded first it would bind the system runtime, torch would then bring a second
    # one, and the two cannot both own the device.  Importing torch first (when it is installed)
    # lets the dynamic loader resolve libamdhip64.so.* to the copy already in the process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.bv_version.restype = C.c_char_p
    L.bv_version.argtypes = []
    L.bv_min_af.restype = C.c_double
    L.bv_min_af.argtypes = [C.c_uint32, C.c_float]
    L.bv_engine_create.restype = C.c_int
    L.bv_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(C.c_void_p)]
    L.bv_engine_destroy.restype = C.c_int
    L.bv_engine_destroy.argtypes = [C.c_void_p]
    L.bv_engine_submit.restype = C.c_int
    L.bv_engine_submit.argtypes = [C.c_void_p, C.POINTER(Slab), C.c_void_p, C.c_void_p, C.c_void_p]
    L.bv_engine_submit_many.restype = C.c_int
    L.bv_engine_submit_many.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Slab), C.POINTER(C.c_void_p), C.c_void_p]
    L.bv_engine_submit_many_g.restype = C.c_int
    L.bv_engine_submit_many_g.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Slab), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]
    L.bv_engine_join.restype = C.c_int
    L.bv_engine_join.argtypes = [C.c_void_p, C.c_void_p]
    L.bv_engine_tiles_begin.restype = C.c_int
    L.bv_engine_tiles_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.bv_engine_tiles_add.restype = C.c_int
    L.bv_engine_tiles_add.argtypes = [C.c_void_p, C.POINTER(Slab), C.c_void_p]
    L.bv_engine_tiles_add_many.restype = C.c_int
    L.bv_engine_tiles_add_many.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Slab), C.c_void_p]
    L.bv_engine_tiles_add_sparse.restype = C.c_int
    L.bv_engine_tiles_add_sparse.argtypes = [C.c_void_p, C.POINTER(SparseTile), C.c_void_p]
    L.bv_sparse_tile_packed_layout.restype = C.c_int
    L.bv_sparse_tile_packed_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.bv_tile_packed_layout.restype = C.c_int
    L.bv_tile_packed_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.bv_engine_tiles_finish.restype = C.c_int
    L.bv_engine_tiles_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.bv_engine_stream.restype = C.c_void_p
    L.bv_engine_stream.argtypes = [C.c_void_p]
    L.bv_engine_wait.restype = C.c_int
    L.bv_engine_wait.argtypes = [C.c_void_p]
    L.bv_engine_kernel_ms.restype = C.c_int
    L.bv_engine_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bv_engine_timing_reset.restype = C.c_int
    L.bv_engine_timing_reset.argtypes = [C.c_void_p]
    L.bv_engine_timing_get.restype = C.c_int
    L.bv_engine_timing_get.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.bv_engine_timing_get_ex.restype = C.c_int
    L.bv_engine_timing_get_ex.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_uint32)]
    L.bv_host_log_probe.restype = C.c_int
    L.bv_host_log_probe.argtypes = [C.POINTER(C.c_double)]
    L.bv_host_log_eval.restype = C.c_double
    L.bv_host_log_eval.argtypes = [C.POINTER(C.c_double), C.c_double]
    L.bv_engine_host_log_exact.restype = C.c_int
    L.bv_engine_host_log_exact.argtypes = [C.c_void_p]
    L.bv_engine_host_log_eval.restype = C.c_int
    L.bv_engine_host_log_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.bv_engine_last_variant_count.restype = C.c_int
    L.bv_engine_last_variant_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.bv_last_error.restype = C.c_char_p
    L.bv_last_error.argtypes = [C.c_void_p]
    L.bv_synth_fill.restype = C.c_int
    L.bv_synth_fill.argtypes = [C.c_int, C.POINTER(SynthParams), C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bv_engine_last_launch_form.restype = C.c_int
    L.bv_engine_last_launch_form.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
    L.bv_device_numa_node.restype = C.c_int
    L.bv_device_numa_node.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.bv_bind_thread_to_device_node.restype = C.c_int
    L.bv_bind_thread_to_device_node.argtypes = [C.c_int]
    _lib = L
    return L
