"""Wall time per launch of the bench's ragged inference batches (128 samples); SOLA_TUNE=key=value,... for A/B runs."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth, _lib
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval(); m.precision = "f16x3"
bs = synth.make_ragged_infer_batches(cfg, 128, 2024)
for tag, b in bs.items():
    videos = [torch.from_numpy(v).cuda() for v in b["videos"]]; texts = [torch.from_numpy(t).cuda() for t in b["texts"]]; sv = b["sample_video"]
    with torch.no_grad():
        for _ in range(3): m.forward_ragged(videos, texts, sv)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.forward_ragged(videos, texts, sv)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    prof = _lib.lib()
    print(f"{tag}: {dt * 1e3:.2f} ms per launch, {128 / dt:.0f} samples/s", flush=True)
