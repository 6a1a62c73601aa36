"""rocprofv3 target: the 16-bit weight-gradient product of one shape on both routes (sola_tune train_tn_tr 1 / 0), five calls each.
usage: gemm_tn_tr_once.py [M N K]"""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import ops, _lib
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 1024, 1024)
a = torch.randn(M, N, device="cuda") * 1e-4; b = torch.randn(M, K, device="cuda")
for v in (1, 0):
    _lib.lib().sola_tune(b"train_tn_tr", v)
    for _ in range(5): ops.gemm_tn_f16(a, b)
torch.cuda.synchronize()
