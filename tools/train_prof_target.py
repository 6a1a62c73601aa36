"""rocprofv3 target: a few training steps at one precision.  usage: train_prof_target.py <batch> <f32|f16x3>"""
import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth
from sola_amd.loss import track_selection_losses
from sola_amd.module import LanguageAlignedTrackSelectionModule
B = int(sys.argv[1]); prec = sys.argv[2]
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
m = m.cuda().train(); m.precision = prec
opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.make_inputs(cfg, B, 64, 32, 16, 1).items()}
for _ in range(6):
    opt.zero_grad(set_to_none=True)
    sm, st = m(inp["object_tokens"], inp["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    loss3 = track_selection_losses(sm, st, inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    loss3[0].backward(); m.clip_grad_norm_(1.0); opt.step()
torch.cuda.synchronize()
