"""One attention shape, a few launches: the target of rocprofv3 --pmc passes (tools/pmc_attn.sh).
usage: attn_pmc_target.py <obj|motion|o2l|obj_bwd|motion_bwd> [B N Tp]   env: SOLA_TUNE="key=val,key=val" """
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops, _lib
kind = sys.argv[1]
B, N, Tp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 64, 4)
for kv in filter(None, os.environ.get("SOLA_TUNE", "").split(",")):
    k, v = kv.split("=")
    _lib.check(_lib.lib().sola_tune(k.encode(), int(v)), "tune")
D, H, Wn = 1024, 8, 48
M = B * N * Tp
q, k, v = (torch.randn(M, D, device="cuda") for _ in range(3))
lk, lv = torch.randn(B * Wn, D, device="cuda"), torch.randn(B * Wn, D, device="cuda")
if kind.endswith("_bwd"):  # the attention backward of the same layout (one-pass kernel by default; SOLA_TUNE=attn_bwd_fused=0 for the two-pass ones)
    geo = {"obj_bwd": (B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)), "motion_bwd": (B * N, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1))}[kind]
    do = torch.randn(M, D, device="cuda")
    o, lse = ops.attention(q, k, v, *geo, return_lse=True)
    for _ in range(5):
        ops.attention_backward(q, k, v, o, do, lse, *geo)
    torch.cuda.synchronize()
    sys.exit(0)
fn = {"obj": lambda: ops.attention(q, k, v, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)),
      "motion": lambda: ops.attention(q, k, v, B * N, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)),
      "o2l": lambda: ops.attention(q, lk, lv, B, H, N * Tp, Wn, 1, (N * Tp, 0, 1), (Wn, 0, 1))}[kind]
for _ in range(5): fn()
torch.cuda.synchronize()
