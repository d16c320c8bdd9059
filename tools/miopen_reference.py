#!/usr/bin/env python
"""Reference point for DESIGN.md: the same 3x3 layer shapes through torch's own convolution (MIOpen, bf16,
channels_last) on the same MI355X -- what a PyTorch-ROCm user gets without libflairhip.  Not used by the
product or the tests.   python tools/miopen_reference.py"""
import torch, time, sys
import torch.nn.functional as F
dev='cuda'
shapes=[("conv64",64,64,128),("conv128",128,128,64),("conv256",256,256,32),("conv512",512,512,16),("dec0c1",768,256,32),("dec2c1",192,64,128)]
for name,cin,cout,H in shapes:
    x=torch.randn(32,cin,H,H,device=dev,dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w=(torch.randn(cout,cin,3,3,device=dev,dtype=torch.bfloat16)/(cin*9)**0.5).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    t0=time.time()
    y=F.conv2d(x,w,padding=1); dy=torch.randn_like(y)
    torch.cuda.synchronize(); t1=time.time()
    def fwd(): return F.conv2d(x,w,padding=1)
    def bwd_in(): return torch.ops.aten.convolution_backward(dy,x,w,None,[1,1],[1,1],[1,1],False,[0,0],1,[True,False,False])
    def bwd_w(): return torch.ops.aten.convolution_backward(dy,x,w,None,[1,1],[1,1],[1,1],False,[0,0],1,[False,True,False])
    flops=2.0*32*H*H*cout*cin*9
    for kn,fn in (("fwd",fwd),("dgrad",bwd_in),("wgrad",bwd_w)):
        with torch.no_grad():
            fn(); torch.cuda.synchronize()
            s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): fn()
            e.record(); torch.cuda.synchronize()
            us=s.elapsed_time(e)/20*1e3
        print(f"{name:8s} {kn:6s} {us:8.1f} us {flops/us/1e6:8.1f} TF  (first call {t1-t0:.1f}s)", flush=True)
