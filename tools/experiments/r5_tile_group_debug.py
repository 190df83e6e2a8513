import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import basevar_amd, oracle
from basevar_amd.synth import make_slab
rng = np.random.default_rng(71)
for it in range(41):
    n = int(rng.choice([8, 37, 64, 300, 2500, 10000, 40000]))
    sites = int(max(16, min(4096, 6_000_000 // n)))
    cov = float(rng.choice([0.02, 0.08, 0.3, 0.9])); qm = float(rng.choice([10.0, 25.0, 33.0]))
    classes = []
    for _ in range(8):
        a = float(rng.choice([0, 0, 0.0005, 0.002, 0.01, 0.05, 0.2, 0.5, 0.95, 1.0])); b = float(rng.choice([0, 0, 0, 0.01, 0.1, 0.3])); classes.append((a, min(b, 1.0 - a)))
    ng = min(int(rng.choice([0, 0, 2, 5, 32, 64 if n <= 60000 else 2])), 32)
    seed = int(rng.integers(1 << 30)); maf_c = float(rng.choice([0.01, 0.001]))
    width = int(rng.choice([7, 64, 200, 1000, max(16, n // 3)]))
slab = make_slab(sites, n, seed=seed, coverage=cov, qual_mean=qm, qual_sd=9.0, qual_min=1, qual_max=60, n_groups=ng, class_af=classes, ref_n_frac=0.03)
res = oracle.Restatement(); maf = res.min_af(n, maf_c)
exp, gexp = (oracle.Reference() if oracle.ref_available() else res).run(slab, maf, n_threads=8)
for flags, w in ((8, width), (8, n), (0, width)):
    eng = basevar_amd.BaseTypeEngine(sites, maf, flags=flags)
    got = eng.lrt_tiles(slab, w)
    rows = eng.lrt(slab)
    eng.close()
    var = (exp["status"] & 2) != 0
    nbad = 0
    for i in np.nonzero(var)[0]:
        for g in range(ng):
            a, b, r = got.groups[i][g], gexp[i][g], rows.groups[i][g]
            if a["n_alt"] != b["n_alt"] or a["alt"].tolist() != b["alt"].tolist():
                nbad += 1
                if nbad <= 6:
                    idx = np.nonzero((slab["group_id"][:n] == g) & (slab["base_strand"][i][:n] < 8))[0]
                    print("flags %d width %d site %d ref %d group %d depth %d: tiles n_alt %d alt %s | reference n_alt %d alt %s | rows n_alt %d alt %s | cells %s" % (
                        flags, w, i, slab["ref_base"][i], g, b["total_depth"], a["n_alt"], a["alt"].tolist(), b["n_alt"], b["alt"].tolist(), r["n_alt"], r["alt"].tolist(),
                        [(int(j), int(slab["base_strand"][i][j]), int(slab["qual"][i][j])) for j in idx]))
    print("flags %d width %d: %d (site, group) calls differ from the reference" % (flags, w, nbad))
