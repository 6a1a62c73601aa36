"""What the in-library HIP-event profiler (two event records per kernel launch; bench.py's timed region runs with it: the roofline's
kernel times come from there) costs the headline step: the same 256-sample step with the profiler on and off, interleaved."""
import gc, sys, time, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import _lib, ops, synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
m = m.cuda().eval()
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, 256, 64, 32, 16, 1000).items()}
def step():
    with torch.no_grad():
        sm, st = m(inp["object_tokens"], inp["lang_tokens"])
        track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], m.negative_token.weight, 1.5, 0.07, 0.3)
        ops.select(sm, 0.5)
for _ in range(3): step()
gc.collect(); gc.disable()
res = {0: [], 1: [], 2: []}
ROOF = ["gemm128", "gemm64", "gemm_split", "gemm_split256", "gemm_split256_gn", "attn"]  # what bench.py's timed region brackets
gemm_us = {1: [], 2: []}
for rnd in range(6):
    for on in (1, 2, 0):
        _lib.profile_enable(bool(on), categories=ROOF if on == 2 else None); _lib.profile_read(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); res[on].append((time.perf_counter() - t0) / 20 * 1e3)
        pr = _lib.profile_read(reset=True)
        if on: gemm_us[on].append(1e3 * pr["gemm_split256"]["ms"] / max(1, pr["gemm_split256"]["launches"]))
_lib.profile_enable(False)
print(f"ms per step, every launch timed: min {min(res[1]):.3f} median {sorted(res[1])[3]:.3f}   GEMM + attention launches timed: min {min(res[2]):.3f} "
      f"median {sorted(res[2])[3]:.3f}   profiler off: min {min(res[0]):.3f} median {sorted(res[0])[3]:.3f}   "
      f"overhead {100 * (min(res[1]) / min(res[0]) - 1):.2f} % / {100 * (min(res[2]) / min(res[0]) - 1):.2f} %")
print(f"average GEMM launch (us): every launch timed {min(gemm_us[1]):.1f}-{max(gemm_us[1]):.1f}, GEMM + attention only {min(gemm_us[2]):.1f}-{max(gemm_us[2]):.1f}")
