import sys, torch
sys.path.insert(0, "/root/repo")
from sola_amd import synth
from sola_amd.loss import track_selection_losses_ragged
from sola_amd.module import LanguageAlignedTrackSelectionModule, collate_ragged
cfg = synth.DEFAULT_MODEL_CFG
for cap in (None, 256 << 20, 32 << 20, 0):
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
    m = m.cuda().train(); m.precision = "bf16"
    if cap is not None:
        m.x16_arena_max_bytes = cap
    smp = synth.make_ragged_samples(cfg, 24, 2024, "cuda")
    objs, langs = collate_ragged([x["obj"] for x in smp]), collate_ragged([x["lang"] for x in smp])
    labels = torch.cat([x["labels"] for x in smp]); pos = torch.stack([x["pos"] for x in smp])
    out = []
    for it in range(3):
        for p in m.parameters(): p.grad = None
        torch.manual_seed(5)
        m.forward_ragged(objs, langs, differentiable=True)
        flat, tok, offs, counts = m.last_ragged
        loss = track_selection_losses_ragged(flat, tok, labels, pos, m.negative_token.weight, offs, counts, 1.5, 0.07, 0.3)
        loss[:, 0].mean().backward()
        g = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
        out.append((float(loss[:, 0].mean()), float(g)))
    print("arena cap", cap, "arena bytes", m.x16_arena_bytes(), out)
