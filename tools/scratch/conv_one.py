"""Lab: one conv shape through the trace build (IFX_HIP_LIB=inferix_amd/libinferix_hip_convtrace.so): H W T [same_frame] [planar]"""
import os,sys
sys.path.insert(0,os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from inferix_amd import hip_ops as ops
h,w,t=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
same=len(sys.argv)>4 and sys.argv[4]=="1"
planar=len(sys.argv)>5 and sys.argv[5]=="1"
c=int(os.environ.get("C","96"))
dev="cuda"; g=torch.Generator(device=dev).manual_seed(0)
rnd=lambda *s: torch.randn(*s,generator=g,device=dev).to(torch.bfloat16)
ring=rnd(t+2,h,w,c); wt=(rnd(27,c//32,c,32)*(27*c)**-0.5).contiguous(); b=rnd(c); res=rnd(t,h,w,c); y=torch.empty(t,h,w,c,dtype=torch.bfloat16,device=dev)
if planar: ring=ops.to_planar(ring)
slots=[0]*(t+2) if same else list(range(t+2))
for _ in range(2):
    ops.conv3d_cl(ring,slots,wt,b,kt=3,ks=3,y=y,out_slots=[0]*t if same else list(range(t)),residual=res[:1].expand(t,h,w,c).contiguous() if False else res)
torch.cuda.synchronize()
