import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops
M,N,K=65536,1024,1024
def t(a,w,n=30):
    for _ in range(5): ops.gemm_nt(a,w)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); [ops.gemm_nt(a,w) for _ in range(n)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
a=torch.randn(M,K,device="cuda"); w=torch.randn(N,K,device="cuda")*0.03
z=torch.zeros(M,K,device="cuda"); zw=torch.zeros(N,K,device="cuda")
for rep in range(3):
    tr=t(a,w); tz=t(z,zw)
    print(f"exact-f32 GEMM {M}x{N}x{K}: random operands {tr:.0f} us ({2*M*N*K/tr/1e6:.1f} TFLOP/s = {2*M*N*K/tr/1e6/157.3:.2f} of 157.3), zero operands {tz:.0f} us ({2*M*N*K/tz/1e6:.1f} TFLOP/s = {2*M*N*K/tz/1e6/157.3:.2f})", flush=True)
