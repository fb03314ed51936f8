import torch, time
dev="cuda"
x=torch.randn(16000,256,device=dev,dtype=torch.bfloat16); w=torch.randn(1024,256,device=dev,dtype=torch.bfloat16)*0.05; b=torch.randn(1024,device=dev,dtype=torch.bfloat16)
ref=torch.nn.functional.gelu(torch.addmm(b,x,w.t()))
try:
    y=torch._addmm_activation(b,x,w.t(),use_gelu=True)
    print("addmm_activation ok; max diff vs erf-gelu:", (y.float()-ref.float()).abs().max().item())
    def t(fn,n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
    print("fused us", t(lambda: torch._addmm_activation(b,x,w.t(),use_gelu=True)), "unfused us", t(lambda: torch.nn.functional.gelu(torch.addmm(b,x,w.t()))))
except Exception as e:
    print("ERR", e)
