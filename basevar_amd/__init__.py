"""basevar_amd -- MI355X-native per-site basetype likelihood engine (hot path of
ShujiaHuang/basevar's `basevar basetype` caller) behind a plain C ABI.

Product = basevar_amd/lib/libbasevar_amd.so (hand-written HIP for gfx950, see csrc/);
this package is the thin host-side mirror of the reference's per-site interface.
"""
from ._capi import (BV_CELL_DEL, BV_CELL_INS, BV_CELL_N, BV_CELL_NOCALL, BV_CELL_REV, BV_SITE_BAD_QUAL,
                    BV_SITE_COVERED, BV_SITE_RANKSUM, BV_SITE_SOR_OVERFLOW, BV_SITE_VARIANT, BV_SITE_ZERO_FREQ,
                    GROUP_DTYPE, SITE_DTYPE)
from .engine import BaseTypeBatch, BaseTypeEngine, min_af, synth_fill, tile_packed_layout

__all__ = ["BaseTypeEngine", "BaseTypeBatch", "min_af", "synth_fill", "tile_packed_layout", "SITE_DTYPE", "GROUP_DTYPE"]
