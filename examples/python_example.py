#!/usr/bin/env python3
"""Minimal use of the engine from Python (needs an MI355X): three hand-made sites, 32 samples each.

    python examples/python_example.py

The planes are the slab of include/basevar_amd.h: one row per site, one column per sample;
base_strand = base code 0..3 (A,C,G,T) | 4 if the read maps to the reverse strand, 8 = no call ('N'), 9 / 10 = an
insertion / deletion token; qual = phred; mapq; rpr = position of the base on its read (1-based)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basevar_amd  # noqa: E402

A, C, G, T, REV, N = 0, 1, 2, 3, 4, 8
n = 32
bs = np.full((3, n), N, np.uint8)
q = np.zeros((3, n), np.uint8)
mq = np.zeros((3, n), np.uint8)
rp = np.zeros((3, n), np.uint16)
# site 0: reference G, every covered sample reads G            -> no variant, one CVG row
# site 1: reference A, 12 x A and 8 x C (both strands)         -> SNV A>C
# site 2: reference T, 9 x G only                              -> hom-alt
cells = {0: [G] * 14 + [G | REV] * 10, 1: [A] * 7 + [A | REV] * 5 + [C] * 4 + [C | REV] * 4, 2: [G] * 5 + [G | REV] * 4}
for s, row in cells.items():
    bs[s, :len(row)] = row
    q[s, :len(row)] = 30 + np.arange(len(row)) % 8
    mq[s, :len(row)] = 60
    rp[s, :len(row)] = 1 + (np.arange(len(row)) * 7) % 100
slab = {"base_strand": bs, "qual": q, "mapq": mq, "rpr": rp, "ref_base": np.array([G, A, T], np.uint8), "n_samples": n}

eng = basevar_amd.BaseTypeEngine(max_sites=3, min_af_value=basevar_amd.min_af(n, 0.01))
bt = eng.lrt(slab)                     # BaseType(...).lrt() + strand_bias + rank sums for the whole batch
eng.close()
for i in range(3):
    r = bt.sites[i]
    print("site %d: depth A,C,G,T = %s  total %d  alts %s  AF %s  QUAL %.2f  FS %.3f  SOR %.3f" % (
        i, list(map(int, r["depth"])), bt.get_total_depth(i), bt.get_alt_bases(i),
        ["%.4f" % bt.get_lrt_af(i, b) for b in bt.get_alt_bases(i)], bt.get_var_qual(i), r["cvg_fs"], r["cvg_sor"]))
assert bt.get_alt_bases(0) == [] and bt.get_alt_bases(1) == ["C"] and bt.get_alt_bases(2) == ["G"]
