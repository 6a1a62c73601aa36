"""rocprofv3 target: the bench's ragged inference batches (128 samples, one / four expressions per video), a few launches each.
usage: ragged_infer_target.py [one|four] [precision]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
tag = {"one": "one_expression_per_video", "four": "four_expressions_per_video"}[sys.argv[1] if len(sys.argv) > 1 else "one"]
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().eval(); m.precision = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
b = synth.make_ragged_infer_batches(cfg, 128, 2024)[tag]
videos = [torch.from_numpy(v).cuda() for v in b["videos"]]
texts = [torch.from_numpy(t).cuda() for t in b["texts"]]
sv = b["sample_video"]
with torch.no_grad():
    for _ in range(6): m.forward_ragged(videos, texts, sv)
torch.cuda.synchronize()
