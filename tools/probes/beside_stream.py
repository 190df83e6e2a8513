# Does a tiny kernel on another stream get through while the short-row streaming kernel runs?  (round 3 diagnosis)
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, basevar_amd
N = 10000; B = 100000; pitch = 10240
dev = torch.device("cuda", 0)
bs = torch.empty((B, pitch), dtype=torch.uint8, device=dev); q = torch.empty_like(bs); ref = torch.empty(B, dtype=torch.uint8, device=dev)
basevar_amd.synth_fill(0, B, N, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), 0, 0, seed=0xBA5E7A7, site_offset=0, coverage=0.08)
eng = basevar_amd.BaseTypeEngine(max_sites=B, min_af_value=basevar_amd.min_af(N), device=0, flags=1 << 24)
out = torch.zeros(B * 208, dtype=torch.uint8, device=dev)
x = torch.zeros(64, device=dev)
big = torch.zeros(256 * 256 * 64, device=dev)
sb = torch.cuda.Stream()
se = torch.cuda.ExternalStream(eng.stream_handle(), device=dev)
for which, t in (("tiny elementwise (1 workgroup)", x), ("elementwise over 4 Mi floats (fills the chip)", big)):
    for rep in range(2):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(se):
            e0.record()
        eng.submit_ptrs(B, N, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), out.data_ptr(), 0, 0)
        evs = []
        with torch.cuda.stream(sb):
            for i in range(40):
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record(); t.add_(1.0); b.record()
                evs.append((a, b))
        eng.wait(); torch.cuda.synchronize()
        print(which, "rep", rep, "(start after submit us : duration us)", " ".join("%d:%d" % (e0.elapsed_time(a) * 1e3, a.elapsed_time(b) * 1e3) for a, b in evs))
