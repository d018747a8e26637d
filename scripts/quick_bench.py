import time, numpy as np, torch
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth
N=1<<20
d = synth.linear_batch(N, 6, 3, 1)
b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
y = torch.from_numpy(np.ascontiguousarray(d["y"].transpose(0,2,1))).cuda()
s = torch.cuda.ExternalStream(b.stream())
for fused,T in ((False,1),(True,10)):
    for _ in range(3): 
        b.update_dev(y[0].data_ptr(), N) if not fused else None
    b.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    K=20
    yy = y[0:1].repeat(T,1,1).contiguous()
    e0.record(s)
    for _ in range(K):
        if fused: b.update_steps_dev(yy.data_ptr(), N, T)
        else: b.update_dev(y[0].data_ptr(), N)
    e1.record(s); b.synchronize()
    ms=e0.elapsed_time(e1)/K
    print("fused" if fused else "single", T, "ms/launch", ms, "steps/s", N*T/(ms*1e-3), "GB/s(1488)", N*T*1488/(ms*1e-3)/1e9)
