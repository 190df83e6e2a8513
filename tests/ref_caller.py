"""ctypes face of oracle/_ref/libbvcaller.so: the reference's own per-position caller (`_basevar_caller`,
/root/reference/src/basetype_caller.cpp:667-762, compiled where it lies), batchfile rows in -> the CVG / VCF bytes it writes out.
Test infrastructure; oracle/ref_caller_driver.cpp says exactly what is and what is not the reference in that library."""
import ctypes as C
import os

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libbvcaller.so")


def available():
    return os.path.exists(LIB)


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.bvref_caller_position.restype = C.c_int
        _lib.bvref_cvg_header.restype = C.c_void_p
        _lib.bvref_caller_free.argtypes = [C.c_void_p]
    return _lib


def cvg_header():
    l = _load()
    p = l.bvref_cvg_header()
    s = C.string_at(p).decode()
    l.bvref_caller_free(p)
    return s


def call_position(rows, n_sample, min_af, groups=None):
    """rows: one batchfile row per file for ONE position (no newline); groups: {name: [sample indices]}.
    Returns (variant, vcf_text, cvg_text); raises RuntimeError with the reference's message where the reference throws."""
    l = _load()
    groups = groups or {}
    names = sorted(groups)
    arr = (C.c_char_p * len(rows))(*[r.encode() for r in rows])
    gn = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
    off, idx = [0], []
    for n in names:
        idx += list(groups[n])
        off.append(len(idx))
    off_a = (C.c_size_t * len(off))(*off)
    idx_a = (C.c_size_t * max(1, len(idx)))(*idx)
    vcf, cvg = C.c_void_p(), C.c_void_p()
    vl, cl = C.c_size_t(), C.c_size_t()
    err = C.create_string_buffer(4096)
    rc = l.bvref_caller_position(arr, len(rows), gn, off_a, idx_a, len(names), C.c_double(min_af), C.c_size_t(n_sample),
                                 C.byref(vcf), C.byref(vl), C.byref(cvg), C.byref(cl), err, C.c_size_t(len(err)))
    if rc < 0:
        raise RuntimeError(err.value.decode(errors="replace"))
    v = C.string_at(vcf, vl.value).decode()
    c = C.string_at(cvg, cl.value).decode()
    l.bvref_caller_free(vcf)
    l.bvref_caller_free(cvg)
    return rc == 1, v, c
