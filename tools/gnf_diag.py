"""Where does the fused conv + GroupNorm epilogue go wrong in a build that fails (e.g. gemm_glds.hip compiled WITH the SLP vectoriser:
SOLA_HIP_LIB=build/libsola_slp.so)?  One fused forward against the unfused launches; the wrong elements of the three taps by row within the
16-row strip, strip index within the wave tile, wave row / column, column within the wave's 64, and GroupNorm instance."""
import sys, collections
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import synth, _lib
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = "f16x3"
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 128, 64, 32, 16, 31).items()}
def unsplit(t):
    h = t.view(torch.float16).reshape(t.shape[0], -1, 2, 8).float()
    return (h[:, :, 0, :] + h[:, :, 1, :]).reshape(t.shape[0], -1)
def run(fuse):
    _lib.lib().sola_tune(b"gemm_gn_fuse", fuse)
    with torch.no_grad(): m(inp["object_tokens"], inp["lang_tokens"])
    torch.cuda.synchronize()
    return {nm: unsplit(m.workspace_tap(nm)).clone() for nm in ("act0", "act1", "act2")}
ref = run(0)
for rep in range(2):
    got = run(1)
    for nm, gnt in (("act0", 16), ("act1", 8), ("act2", 4)):
        d = (got[nm] - ref[nm]).abs()
        bad = (d > 1e-3).nonzero()
        print(f"run {rep} {nm} (instances of {gnt} rows): {bad.shape[0]} wrong of {d.numel()}, worst {float(d.max()):.3f}", flush=True)
        if bad.shape[0] == 0: continue
        r, c = bad[:, 0].cpu(), bad[:, 1].cpu()
        def hist(name, x, top=8):
            cnt = collections.Counter(x.tolist()); print(f"    {name}: " + ", ".join(f"{k}:{v}" for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:top]) + f"  ({len(cnt)} distinct)")
        hist("row % 16 (row in strip)", r % 16); hist("strip in wave tile (row // 16 % 8)", (r // 16) % 8); hist("wave row (row // 128 % 2)", (r // 128) % 2)
        hist("tile row (row // 256)", r // 256, 6); hist("col % 64", c % 64, 10); hist("wave col (col // 64 % 4)", (c // 64) % 4); hist("col % 4", c % 4)
        hist("instance (row // gnt) % (16 // gnt)", (r // gnt) % (16 // gnt))
        # are whole (instance, wave column) units wrong?
        unit = (r // gnt) * 1000 + c // 64
        cnt = collections.Counter(unit.tolist()); sizes = collections.Counter(cnt.values())
        print(f"    wrong elements per (instance, 64-column group): " + ", ".join(f"{k} elems x {v}" for k, v in sorted(sizes.items(), key=lambda kv: -kv[1])[:6]) + f"  (full unit = {gnt * 64})")
