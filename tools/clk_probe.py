import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from mamba_asr_amd import ops, _native
dev="cuda"; dt=torch.bfloat16
b,e,l,n=16,512,1000,16
g=torch.Generator(device=dev).manual_seed(0)
ucl=torch.randn(b,l,e,device=dev,generator=g).to(dt); dcl=(torch.randn(b,l,e,device=dev,generator=g)*0.5).to(dt)
xz=torch.randn(b,l,2*e,device=dev,generator=g).to(dt)
A=-torch.exp(torch.randn(e,n,device=dev,generator=g)*0.3); D=torch.randn(e,device=dev,generator=g); bias=torch.randn(e,device=dev,generator=g)-1
Bcl,Ccl=ops.alloc_bc(16,b,l,dev),ops.alloc_bc(16,b,l,dev); Bcl.normal_(generator=g); Ccl.normal_(generator=g)
ycat=torch.empty(b,l,2*e,device=dev,dtype=dt)
dirs=[dict(u=ucl,delta=dcl,A=A,B=Bcl,C=Ccl,D=D,delta_bias=bias,out=ycat[:,:,i*e:(i+1)*e],reverse=bool(i)) for i in range(2)]
def t(fn,iters):
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters
import ctypes
h=ctypes.CDLL(_native.LIB_PATH)
for abl,name in ((0,"baseline"),(1,"no barrier"),(2,"no exp"),(3,"no global ld/st"),(4,"no ybuf exchange")):
    h.cm_debug_set(abl)
    ms=t(lambda: ops.scan_cl_fwd(dirs,z=xz[:,:,e:],split=8),50)
    print(f"abl {abl} ({name}): {ms*1e3:.1f} us")
h.cm_debug_set(5)
ms=t(lambda: ops.scan_cl_fwd(dirs,z=xz[:,:,e:],split=8),3)
buf=(ctypes.c_ulonglong*8)(); h.cm_debug_read_stamps(buf)
n=max(1,buf[4]); print("stamps per iteration (cycles): loads-issue+ypart %.0f | compute+ywrite %.0f | gate+produce %.0f | barrier %.0f | iters %d" % (buf[0]/n,buf[1]/n,buf[2]/n,buf[3]/n,buf[4]))
h.cm_debug_set(0)
