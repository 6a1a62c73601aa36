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
for (M,N,K) in [(65536,1024,1024),(65536,1024,3072),(40930,1024,1024),(16384,1024,1024),(262144,512,768)]:
    a=torch.randn(M,K,device="cuda"); w=torch.randn(N,K,device="cuda")*0.03; b=torch.randn(N,device="cuda"); r=torch.randn(M,N,device="cuda")
    out={}
    for v in (0,1):
        lib.sola_tune(b"gemm_f32_nw8", v)
        out[v]=(t(lambda: ops.gemm_nt(a,w,b,r)), ops.gemm_nt(a,w,b,r))
    lib.sola_tune(b"gemm_f32_nw8", 1)
    same=torch.equal(out[0][1],out[1][1])
    print(f"M={M} N={N} K={K}: four waves {out[0][0]:.0f} us ({2*M*N*K/out[0][0]/1e6:.1f} TF) -> eight waves {out[1][0]:.0f} us ({2*M*N*K/out[1][0]/1e6:.1f} TF = {2*M*N*K/out[1][0]/1e6/157.3:.2f} of peak); bit-identical {same}", flush=True)
