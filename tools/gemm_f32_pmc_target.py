"""One exact-f32 GEMM shape, a few launches: the target of rocprofv3 --pmc passes (tools/pmc_gemm_f32.sh).  usage: gemm_f32_pmc_target.py [M N K] [variant 0|1]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 1024, 1024)
if len(sys.argv) > 4: _lib.lib().sola_tune(b"gemm_variant", int(sys.argv[4]))
a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
for _ in range(5): ops.gemm_nt(a, w, b)
torch.cuda.synchronize()
