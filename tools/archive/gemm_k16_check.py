"""Experiment "gemm_k16" (256x128 tiles, 16-deep k-tiles, two four-wave blocks per CU): bit-identity against the default kernels over
residual / output formats / ragged edges, and of a whole default-precision forward (conv GEMMs included)."""
import sys, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import ops, _lib, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
lib = _lib.lib()
torch.manual_seed(0)
bad = 0
for (M, N, K, res, osp) in [(65536, 1024, 1024, 0, 0), (65536, 1024, 1024, 1, 0), (65536, 1024, 1024, 1, 1), (65536, 1024, 1024, 0, 1),
                            (65536 - 77, 1024, 1024, 1, 0), (65536, 1024 - 8, 1024, 0, 1), (131072, 512, 768, 0, 0), (65536, 1024, 32, 0, 0),
                            (65536, 1024, 96, 1, 1)]:
    a = ops.cast_sp16(torch.randn(M, K, device="cuda")); w = ops.cast_sp16(torch.randn(N, K, device="cuda") * 0.03, 64.0)
    b = torch.randn(N, device="cuda"); r = ops.cast_sp16(torch.randn(M, N, device="cuda")) if res else None
    outs = []
    for v in (0, 1):
        lib.sola_tune(b"gemm_k16", v)
        outs.append(ops.gemm_nt_split(a, w, b, r, True, 1 / 64, bool(osp)).clone())
    same = bool(torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)))
    bad += not same
    print(f"M={M} N={N} K={K} residual={res} out_split={osp}: bit-identical {same}", flush=True)
cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval()
inp = synth.make_inputs(cfg, 64, 64, 32, 16, 0)
ot, lt = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
res = []
for v in (0, 1):
    lib.sola_tune(b"gemm_k16", v)
    with torch.no_grad():
        sm, st = m(ot, lt)
    res.append((sm.clone(), st.clone()))
same = bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]))
bad += not same
print("forward (B=64, N=64, T=32, default precision): bit-identical", same)
lib.sola_tune(b"gemm_k16", 0)
sys.exit(1 if bad else 0)
