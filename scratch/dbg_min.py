import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from bench import synth_mels
from tacotron2_amd import filters
from tacotron2_amd.plan import Plan
N,H,T,B=2048,256,861,2
p=Plan(N,H,T,B); 
fb=torch.from_numpy(filters.mel(22050,N,80,125,7600)).cuda()
mel=torch.from_numpy(synth_mels(B,T,1234)).cuda()
mag=p.mel_to_mag(fb,mel)
x0=p.randn((B,T*H+N),seed=42)
for K in (0,1,2,5):
    x=p.run(mag,x0,K)
    S=p.stft(x)
    r2=(S.real**2+S.imag**2)
    print(K, 'min r2', float(r2.min()), 'frames with bin<1e-30:', int((r2.min(dim=2).values<1e-30).sum()), 'of', B*T, ' n bins <1e-30', int((r2<1e-30).sum()))
    if K==1:
        idx=(r2<1e-30).nonzero()[:10]; print(idx)
print('mag zeros per frame', int((mag[0,0]==0).sum()))
