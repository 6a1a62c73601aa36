"""Where a ragged scoring call of inference.py spends its host time (128 samples of the synthetic MeViS-like mix)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops, synth
from sola_amd.data import SyntheticTracks, RaggedBatcher, DevicePrefetcher
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder
cfg = synth.DEFAULT_MODEL_CFG
dev = torch.device("cuda", 0)
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.to(dev).eval(); m.precision = os.environ.get("SOLA_PRECISION", "f16x3")
text = TextEncoder("none", cfg["lang_token_dim"], dev, allow_standin=True)
ds = SyntheticTracks(n_samples=512, token_dim=256, seed=2, with_labels=False, per_video=4, ragged=True)
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + t - t0; return t
n = 0
with torch.no_grad():
    for batch in DevicePrefetcher(RaggedBatcher(ds, range(512), 128), dev):
        t = time.perf_counter()
        texts, _ = text.encode_ragged([s["expression"] for s in batch["samples"]]); t = tick("text", t)
        m.forward_ragged(batch["videos"], texts, batch["sample_video"]); t = tick("forward_ragged", t)
        flat, _tok, _offs, counts = m.last_ragged
        _p, pred = ops.select(flat, 0.5); pred = pred.cpu().numpy(); t = tick("select+copy", t)
        n += 1
print({k: round(1e3 * v / n, 2) for k, v in acc.items()}, "ms per 128-sample call")
