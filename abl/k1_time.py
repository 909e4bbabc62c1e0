import torch, numpy as np, sys
sys.path.insert(0, '/root/repo')
from tacotron2_amd.plan import Plan
from tacotron2_amd import filters
N,H,T,B=2048,256,861,64
plan=Plan(N,H,T,B)
fb=torch.from_numpy(filters.mel(22050,N,80,125,7600)).cuda()
mel=torch.rand((B,80,T),device='cuda')
x0=plan.randn((B,plan.signal_length(T)),seed=1)
mag=torch.empty((B,T,N//2+1),device='cuda'); x=torch.empty_like(x0)
def t(f,n=20):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for _ in range(3):
    print("mel_to_mag us", t(lambda: plan.mel_to_mag(fb,mel,out=mag)))
    print("mel_to_mag+run(1) us", t(lambda: (plan.mel_to_mag(fb,mel,out=mag), plan.run(mag,x0,1,out=x))))
    print("mel_run(1) us", t(lambda: plan.mel_run(fb,mel,x0,1,out=x)))
    print("run(1) us", t(lambda: plan.run(mag,x0,1,out=x)))
