"""Does running two half-batches on two HIP streams (each its own context/workspace) beat one full batch on one stream?
Kernels of the two streams can overlap: the memory-bound GEMM epilogues / GroupNorm / attention of one with the MFMA loops of
the other.  Prints samples/s for 1 stream x B and 2 streams x B/2, for the 128x128-only and the auto GEMM shapes."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import _lib, synth, ops
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(cfg, 42)
def make():
    m = LanguageAlignedTrackSelectionModule(cfg); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.to(dev).eval(); m.precision = "f16x3"; return m
ms = [make(), make()]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inp = synth.make_inputs(cfg, B, 64, 32, 16, seed=1000)
obj = torch.from_numpy(inp["object_tokens"]).to(dev); lang = torch.from_numpy(inp["lang_tokens"]).to(dev)
labels = torch.from_numpy(inp["labels"]).to(dev); pos = torch.from_numpy(inp["pos_tokens"]).to(dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def step(nstreams):
    h = B // nstreams
    outs = []
    for i in range(nstreams):
        with torch.cuda.stream(streams[i]), torch.no_grad():
            m = ms[i]; sl = slice(i * h, (i + 1) * h)
            sm, st = m(obj[sl], lang[sl])
            l3 = track_selection_losses(sm, st, labels[sl], pos[sl], m.negative_token.weight, 1.5, 0.07, 0.3)
            outs.append((l3, ops.select(sm, 0.5)[1]))
    return outs
for glds in (3, 1):
    _lib.lib().sola_tune(b"gemm_glds", glds)
    for ns in (1, 2):
        for _ in range(3): step(ns)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 10
        for _ in range(n): step(ns)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(f"gemm_glds={glds} streams={ns} batch/stream={B // ns}: {B * n / el:.0f} samples/s, {1e3 * el / n:.2f} ms/step")
