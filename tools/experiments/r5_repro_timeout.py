"""Reproduces the hand-off time-out the round-5 campaign found (seed 64, slab 38: 1,464 sites x 4,097 samples on TWO workgroups,
BV_FLAG_GRID_LIMIT(2): 732 sites per workgroup, far more candidates and variant sites than the LDS queues hold)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import basevar_amd, oracle
from basevar_amd.synth import make_slab
classes = [(0.002, 0.01), (0.2, 0.0), (0.0, 0.0), (0.5, 0.0), (0.0, 0.01), (1.0, 0.0), (0.002, 0.0), (0.05, 0.01)]
slab = make_slab(1464, 4097, seed=257428233, coverage=0.02, qual_mean=25.0, qual_sd=9.0, qual_min=1, qual_max=60, n_groups=0,
                 class_af=classes, ref_n_frac=0.03)
maf = oracle.Restatement().min_af(4097, 0.01)
for flags in [int(a, 0) for a in sys.argv[1:]] or [2 << 16]:
    for rep in range(3):
        eng = basevar_amd.BaseTypeEngine(1464, maf, flags=flags)
        t0 = time.time()
        try:
            got = eng.lrt(slab)
            print("flags %#x rep %d: ok, %d variant sites, %.3f s" % (flags, rep, got.n_variant, time.time() - t0), flush=True)
        except RuntimeError as ex:
            print("flags %#x rep %d: %s (%.3f s)" % (flags, rep, ex, time.time() - t0), flush=True)
        eng.close()
