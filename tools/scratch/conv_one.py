import os,sys
sys.path.insert(0,'/root/repo')
import torch
from inferix_amd import hip_ops as ops
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
rnd=lambda *s: torch.randn(*s,generator=g,device=dev).to(torch.bfloat16)
c,h,w,t=96,480,832,12
ring=rnd(t+2,h,w,c); wt=(rnd(27,c//32,c,32)*(27*c)**-0.5).contiguous(); b=rnd(c); res=rnd(t,h,w,c); y=torch.empty(t,h,w,c,dtype=torch.bfloat16,device=dev)
for _ in range(2):
    ops.conv3d_cl(ring,list(range(t+2)),wt,b,kt=3,ks=3,y=y,out_slots=list(range(t)),residual=res)
torch.cuda.synchronize()
