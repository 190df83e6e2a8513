import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, basevar_amd as bv
from basevar_amd.synth import make_slab
n, groups = 60000, 3
sl = make_slab(200, n, seed=302, coverage=0.09, class_af=[(0.0, 0.0), (0.3, 0.0), (0.2, 0.1)])
maf = bv.min_af(n); dev = torch.device("cuda", 0)
rng = np.random.default_rng(n + groups)
g = rng.integers(0, groups + 1, size=sl["pitch"]).astype(np.uint8); g[g == groups] = 255
gid_t = torch.from_numpy(g).to(dev)
rec, grec = bv.SITE_DTYPE.itemsize, bv.GROUP_DTYPE.itemsize
t = [torch.from_numpy(np.ascontiguousarray(sl[k])).to(dev) for k in ("base_strand", "qual", "ref_base", "mapq")]
t.append(torch.from_numpy(np.ascontiguousarray(sl["rpr"]).view(np.int16)).to(dev))
def run(eng, S):
    out = torch.zeros(S * rec, dtype=torch.uint8, device=dev); gout = torch.zeros(S * groups * grec, dtype=torch.uint8, device=dev)
    eng.submit_ptrs(S, n, sl["pitch"], t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), group_id=gid_t.data_ptr(), n_groups=groups, gout=gout.data_ptr())
    eng.wait()
    return gout.cpu().numpy().view(bv.GROUP_DTYPE).copy()
def diff(a, b):
    m = min(len(a), len(b))
    return sum(x.tobytes() != y.tobytes() for x, y in zip(a[:m], b[:m]))
e1 = bv.BaseTypeEngine(max_sites=1300, min_af_value=maf, device=0)
a1 = run(e1, 200); a2 = run(e1, 200); a3 = run(e1, 96)
e2 = bv.BaseTypeEngine(max_sites=1300, min_af_value=maf, device=0)
b1 = run(e2, 200); b3 = run(e2, 96)
print("same engine, same submit twice:", diff(a1, a2), "| 200 sites vs the first 96 of them:", diff(a1, a3), "| other engine:", diff(a1, b1), diff(a3, b3))
e3 = bv.BaseTypeEngine(max_sites=1300, min_af_value=maf, device=0)
c3 = run(e3, 96); c1 = run(e3, 200)
print("fresh engine, 96 first then 200:", diff(c3, a3), diff(c1, a1), diff(c3, c1))
