"""Stress for the team form of bv_pass1_kernel: many small long-row launches of random shape, each compared byte for byte with the
plain kernel (flags 1 << 8 = shape 1).  Hand-offs through LDS flags are timing-dependent; a lost update shows as a mismatch or
as the kernel's time-out flag (bv_engine_wait then raises)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
g.build()
import basevar_amd as bv
from basevar_amd.synth import make_slab

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(20261003)
total = 0
for r in range(rounds):
    n = int(rng.choice([49153, 52000, 70001, 100000, 180000]))
    S = int(rng.choice([1, 7, 300, 1023, 1024, 1025, 2500, 5000]))
    cov = float(rng.choice([0.01, 0.06, 0.3]))
    slab = make_slab(S, n, seed=int(rng.integers(1 << 30)), coverage=cov, site_offset=int(rng.integers(100)))
    if S > 10:
        slab["base_strand"][3, :] = 0x08
        slab["base_strand"][5, 50:] = 0x08
    maf = bv.min_af(n)
    e1 = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0)
    e2 = bv.BaseTypeEngine(max_sites=S, min_af_value=maf, device=0, flags=1 << 8)
    for rep in range(3):
        a = e1.lrt(slab)
        b = e2.lrt(slab)
        assert a.sites.tobytes() == b.sites.tobytes(), (r, rep, n, S, cov)
        assert a.n_variant == b.n_variant
        total += S
    e1.close(); e2.close()
print("team form vs plain kernel: %d launches, %d site records, all byte-identical" % (rounds * 3, total))
