import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, basevar_amd as bv
from basevar_amd.synth import make_slab
n, groups = 60000, 3
sizes = [96, 17, 200, 64]
slabs = [make_slab(s, n, seed=300 + k, coverage=(0.05 + 0.02 * (k % 3)), class_af=[(0.0, 0.0), (0.3, 0.0), (0.2, 0.1)]) for k, s in enumerate(sizes)]
maf = bv.min_af(n); dev = torch.device("cuda", 0)
rng = np.random.default_rng(n + groups)
g = rng.integers(0, groups + 1, size=slabs[0]["pitch"]).astype(np.uint8); g[g == groups] = 255
gid_t = torch.from_numpy(g).to(dev)
rec, grec = bv.SITE_DTYPE.itemsize, bv.GROUP_DTYPE.itemsize
keep, segs, outs, gouts = [], [], [], []
for sl in slabs:
    t = [torch.from_numpy(np.ascontiguousarray(sl[k])).to(dev) for k in ("base_strand", "qual", "ref_base", "mapq")]
    t.append(torch.from_numpy(np.ascontiguousarray(sl["rpr"]).view(np.int16)).to(dev))
    out = torch.zeros(sl["n_sites"] * rec, dtype=torch.uint8, device=dev)
    gout = torch.zeros(sl["n_sites"] * groups * grec, dtype=torch.uint8, device=dev)
    keep.append(t); outs.append(out); gouts.append(gout)
    segs.append((sl["n_sites"], t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr(), t[3].data_ptr(), t[4].data_ptr()))
def singles(eng):
    res = []
    for k, sl in enumerate(slabs):
        gouts[k].zero_()
        eng.submit_ptrs(sl["n_sites"], n, sl["pitch"], segs[k][1], segs[k][2], segs[k][3], segs[k][4], segs[k][5], segs[k][6], group_id=gid_t.data_ptr(), n_groups=groups, gout=gouts[k].data_ptr())
        eng.wait()
        res.append(gouts[k].cpu().numpy().view(bv.GROUP_DTYPE).copy())
    return res
def chained(eng):
    for x in gouts: x.zero_()
    eng.submit_many_ptrs(n, slabs[0]["pitch"], segs, group_id=gid_t.data_ptr(), n_groups=groups, gouts=[x.data_ptr() for x in gouts])
    eng.wait()
    return [x.cpu().numpy().view(bv.GROUP_DTYPE).copy() for x in gouts]
def diff(A, B):
    return [sum(x.tobytes() != y.tobytes() for x, y in zip(a, b)) for a, b in zip(A, B)]
e1 = bv.BaseTypeEngine(max_sites=sum(sizes), min_af_value=maf, device=0)
s_first = singles(e1)
c1 = chained(e1)
s_after = singles(e1)
c2 = chained(e1)
e2 = bv.BaseTypeEngine(max_sites=sum(sizes), min_af_value=maf, device=0)
c_fresh = chained(e2)
s_fresh_after = singles(e2)
print("singles before vs chained:", diff(s_first, c1))
print("singles after  vs chained:", diff(s_after, c1))
print("singles before vs after  :", diff(s_first, s_after))
print("chained twice            :", diff(c1, c2))
print("fresh chained vs chained :", diff(c_fresh, c1), " fresh engine singles after its chained vs singles before:", diff(s_fresh_after, s_first))
