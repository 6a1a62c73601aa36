"""Soak of the persistent split GEMM against the one-tile kernel: random multi-round shapes, every epilogue flavour, results
must be bit-identical launch after launch (a lost DMA wait or strip hazard would show up as a mismatch)."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import ops, _lib
lib = _lib.lib()
def tune(**kv):
    for k, v in kv.items(): lib.sola_tune(k.encode(), int(v))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0 = time.time(); n = 0; bad = 0
tune(gemm_glds_force=1, gemm_glds=4)
while time.time() - t0 < budget:
    M = int(rng.integers(256, 90000)); N = 8 * int(rng.integers(16, 160)); K = 32 * int(rng.integers(2, 40))
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda"); rf = torch.randn(M, N, device="cuda"); rs = ops.cast_sp16(rf)
    for res, rsplit, osp in [(None, False, False), (rf, False, False), (rs, True, False), (None, False, True), (rs, True, True)]:
        tune(gemm_persist=0); ref = ops.gemm_nt_split(a, w, b, res, rsplit, 1 / 64, osp).clone()
        tune(gemm_persist=1)
        for rep in range(3):
            got = ops.gemm_nt_split(a, w, b, res, rsplit, 1 / 64, osp)
            if not torch.equal(got.view(torch.int32), ref.view(torch.int32)):
                bad += 1
                print("MISMATCH", (M, N, K), res is not None, rsplit, osp, float((got - ref).abs().max()), flush=True)
        n += 1
    del a, w, rf, rs
tune(gemm_glds_force=0, gemm_glds=3, gemm_persist=1)
print(f"{n} launch configurations x 3 repeats, {bad} mismatches")
