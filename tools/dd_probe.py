import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_dropout import _train_step
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.SMALL_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda(); shape = (2, 6, 24, 7)
params = dict(m.named_parameters())
gen = torch.Generator(device="cuda").manual_seed(0)
direction = {k: torch.randn(p.shape, device="cuda", generator=gen) * (p.detach().abs().mean() + 1e-3) for k, p in params.items()}
for mode in ("eval", "train"):
    getattr(m, mode)()
    torch.manual_seed(11); _, l0, g = _train_step(m, cfg, *shape, 5)
    analytic = sum(float((g[k].double() * direction[k].double()).sum()) for k in params)
    for eps in (2e-3, 5e-4, 1e-4, 2e-5):
        vals = []
        for sgn in (+1, -1):
            with torch.no_grad():
                for k, p in params.items(): p.add_(sgn * eps * direction[k])
            torch.manual_seed(11); _, l, _ = _train_step(m, cfg, *shape, 5); vals.append(float(l[0]))
            with torch.no_grad():
                for k, p in params.items(): p.sub_(sgn * eps * direction[k])
        print(mode, eps, "numeric", (vals[0] - vals[1]) / (2 * eps), "analytic", analytic)
