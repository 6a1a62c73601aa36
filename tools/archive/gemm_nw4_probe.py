"""Experiment: the persistent split-f16 GEMM on 256x128 tiles with four waves (one per SIMD) against the 256x256 / eight-wave
kernel: bit-identity of the result and time with / without the epilogue (sola_tune gemm_nw4, gemm_ablate)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import ops, _lib
lib = _lib.lib()
KEY = sys.argv[1].encode() if len(sys.argv) > 1 else b"gemm_nw4"
SPLIT = len(sys.argv) > 2 and sys.argv[2] == "split"
for (M, N, K) in [(65536, 1024, 1024), (131072, 512, 768), (65536, 1024, 3072), (16384, 1024, 1024)]:
    x = torch.randn(M, K, device="cuda"); wt = torch.randn(N, K, device="cuda") * 0.03
    a = ops.cast_sp16(x); w = ops.cast_sp16(wt, 64.0); b = torch.randn(N, device="cuda")
    outs = {}
    for nw4 in (0, 1):
        lib.sola_tune(KEY, nw4)
        lib.sola_tune(b"gemm_ablate", 0)
        outs[nw4] = ops.gemm_nt_split(a, w, b, out_scale=1 / 64, out_split=SPLIT).clone()
        row = []
        for ab in (0, 4):
            lib.sola_tune(b"gemm_ablate", ab)
            best = 1e9
            for rnd in range(3):
                ops.gemm_nt_split(a, w, b, out_scale=1 / 64, out_split=SPLIT); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops.gemm_nt_split(a, w, b, out_scale=1 / 64, out_split=SPLIT)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            row.append(best * 1e3)
        print(f"M={M} N={N} K={K} {(KEY.decode() + ' on') if nw4 else '8 waves, 256x256'}: full {row[0]:7.1f} us   no epilogue {row[1]:7.1f} us")
    lib.sola_tune(b"gemm_ablate", 0)
    print("   bit-identical:", bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))), " max |diff|", float((outs[0] - outs[1]).abs().max()))
lib.sola_tune(KEY, 0)
