import sys, time, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth, _lib
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.loss import track_selection_losses
cfg = dict(synth.DEFAULT_MODEL_CFG); cfg["dropout_p"] = 0.0
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().train(); m.attention_dropout_p = 0.0
opt = torch.optim.AdamW(m.parameters(), lr=5e-6)
inp = synth.make_inputs(cfg, B, 64, 32, 16, 1)
c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
def step():
    sm, st = m(c["object_tokens"], c["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    loss3 = track_selection_losses(sm, st, c["labels"], c["pos_tokens"], neg, 1.5, 0.07, 0.3)
    opt.zero_grad(); loss3[0].backward()
    gn = m.get_grad_norm_dict()
    if gn["total_grad_norm"] > 1.0: torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    return loss3
for _ in range(2): l = step()
torch.cuda.synchronize(); _lib.profile_enable(True); _lib.profile_read(True)
t = time.perf_counter(); K = 5
for _ in range(K): l = step()
torch.cuda.synchronize(); el = (time.perf_counter() - t) / K
prof = _lib.profile_read(True)
print(f"B={B} train step {el*1e3:.2f} ms -> {B/el:.1f} samples/s loss {l.tolist()}")
for k, v in prof.items():
    if v["launches"]: print(f"  {k:14s} {v['launches']/K:6.0f} launches {v['ms']/K:8.3f} ms/step  {v['flops']/max(v['ms'],1e-9)/1e9:8.1f} TF/s {v['bytes']/max(v['ms'],1e-9)/1e6:8.1f} GB/s")
