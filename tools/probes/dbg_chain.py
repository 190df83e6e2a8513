import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, basevar_amd as bv
from basevar_amd.synth import make_slab
n, groups = 60000, 3
sizes = [96, 17, 200, 64, 1, 130, 48, 77, 33, 120, 5, 5, 60, 41, 9, 88, 150, 3, 70]
slabs = [make_slab(s, n, seed=300 + k, coverage=(0.05 + 0.02 * (k % 3)), class_af=[(0.0, 0.0), (0.3, 0.0), (0.2, 0.1)]) for k, s in enumerate(sizes)]
maf = bv.min_af(n); dev = torch.device("cuda", 0)
rng = np.random.default_rng(n + groups)
g = rng.integers(0, groups + 1, size=slabs[0]["pitch"]).astype(np.uint8); g[g == groups] = 255
g[: n // 3][g[: n // 3] == 1] = 255
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
eng = bv.BaseTypeEngine(max_sites=sum(sizes), min_af_value=maf, device=0)
eng.submit_many_ptrs(n, slabs[0]["pitch"], segs, group_id=gid_t.data_ptr(), n_groups=groups, gouts=[x.data_ptr() for x in gouts])
eng.wait()
ch = [x.cpu().numpy().view(bv.GROUP_DTYPE).copy() for x in gouts]
eng.close()
eng = bv.BaseTypeEngine(max_sites=sum(sizes), min_af_value=maf, device=0)
import oracle, os
if os.environ.get("DBG_WARM"):
    # one submit of the biggest slab first
    k = 2; sl = slabs[k]
    eng.submit_ptrs(sl["n_sites"], n, sl["pitch"], segs[k][1], segs[k][2], segs[k][3], segs[k][4], segs[k][5], segs[k][6], group_id=gid_t.data_ptr(), n_groups=groups, gout=gouts[k].data_ptr())
    eng.wait()
for k, sl in enumerate(slabs):
    gouts[k].zero_()
    eng.submit_ptrs(sl["n_sites"], n, sl["pitch"], segs[k][1], segs[k][2], segs[k][3], segs[k][4], segs[k][5], segs[k][6], group_id=gid_t.data_ptr(), n_groups=groups, gout=gouts[k].data_ptr())
    eng.wait()
    one = gouts[k].cpu().numpy().view(bv.GROUP_DTYPE)
    bad = np.nonzero([a.tobytes() != b.tobytes() for a, b in zip(one, ch[k])])[0]
    if len(bad): print("slab", k, "differing group records", len(bad), "of", len(one))
    if 0:
        d = dict(sl); d["group_id"] = g; d["n_groups"] = groups
        exp, gexp = oracle.Restatement().run(d, maf)
    for i in bad[:0]:
        print("  oracle af", gexp[i // groups, i % groups]["af"][:2] if gexp.ndim == 2 else gexp[i]["af"][:2])
        print("  rec", i, "site", i // groups, "group", i % groups, "single", one[i]["n_alt"], one[i]["total_depth"], one[i]["af"][:2], "chained", ch[k][i]["n_alt"], ch[k][i]["total_depth"], ch[k][i]["af"][:2], "rel", abs(one[i]["af"][0]-ch[k][i]["af"][0])/max(abs(one[i]["af"][0]),1e-300))
