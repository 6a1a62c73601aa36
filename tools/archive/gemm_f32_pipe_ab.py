import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib=_lib.lib()
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); [fn() for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (M,N,K) in [(65536,1024,1024),(65536,1024,3072),(40930,1024,1024),(262144,512,768)]:
    a=torch.randn(M,K,device="cuda"); w=torch.randn(N,K,device="cuda")*0.03; b=torch.randn(N,device="cuda"); r=torch.randn(M,N,device="cuda")
    line=f"M={M} N={N} K={K}:"
    for v in (0,1):
        lib.sola_tune(b"gemm_variant", v)
        line+=f"  pipe{v} {t(lambda: ops.gemm_nt(a,w,b,r)):.0f} us / no-res {t(lambda: ops.gemm_nt(a,w)):.0f} us"
    lib.sola_tune(b"gemm_variant", -1)
    print(line, flush=True)
