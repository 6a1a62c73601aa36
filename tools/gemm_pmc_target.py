"""One split-f16 GEMM shape, one kernel variant, a few launches: the target of rocprofv3 --pmc passes.
usage: gemm_pmc_target.py <variant 0|1|2> [M N K]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
v = int(sys.argv[1]); M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (16384, 1024, 1024)
x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda"))
_lib.lib().sola_tune(b"gemm_glds", v)
import os
_lib.lib().sola_tune(b"gemm_ablate", int(os.environ.get("SOLA_ABLATE", "0")))
_lib.lib().sola_tune(b"gemm_persist", int(os.environ.get("SOLA_PERSIST", "1")))
_lib.lib().sola_tune(b"gemm_k16", int(os.environ.get("SOLA_K16", "0")))
if os.environ.get("SOLA_NORES"): r = None
for _ in range(5): ops.gemm_nt_split(a, w, b, r, True, 1 / 64)
torch.cuda.synchronize()
