import sys
sys.path.insert(0,".")
import numpy as np, torch, basevar_amd
N=10000; B=25000; pitch=10240
dev=torch.device("cuda",0)
bs=torch.empty((B,pitch),dtype=torch.uint8,device=dev); q=torch.empty_like(bs); ref=torch.empty(B,dtype=torch.uint8,device=dev)
basevar_amd.synth_fill(0,B,N,pitch,bs.data_ptr(),q.data_ptr(),ref.data_ptr(),0,0,seed=0xBA5E7A7,site_offset=0,coverage=0.08)
eng=basevar_amd.BaseTypeEngine(max_sites=B,min_af_value=basevar_amd.min_af(N),device=0,flags=1<<24)
out=torch.zeros(B*208,dtype=torch.uint8,device=dev)
eng.submit_ptrs(B,N,pitch,bs.data_ptr(),q.data_ptr(),ref.data_ptr(),out.data_ptr(),0,0)
eng.wait()
r=out.cpu().numpy().view(basevar_amd.SITE_DTYPE)
for k in (3,4):
    m=r["n_em"]==k
    print("n_em",k,"sites",m.sum(),"em_iters hist",np.bincount(r["em_iters"][m])[:40], "variant", ((r["status"][m]&2)!=0).sum())
    sb=r["cvg_sb"][m]
    n1=sb[:,0]+sb[:,1]; n_1=sb[:,0]+sb[:,2]; n=sb.sum(1)
    tabs=np.minimum(n1,n_1)-np.maximum(0,n1+n_1-n)+1
    print("  cvg fisher tables: mean %.1f p50 %d p90 %d max %d" % (tabs.mean(), np.median(tabs), np.percentile(tabs,90), tabs.max()), "total depth mean", r["total_depth"][m].mean())
