"""Full-model gradients at B=8 N=40: exact-f32 path, split-f16 path and float64 autograd through the oracle (CPU)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as tb
from sola_amd import synth
from oracle import sola_oracle
m, sd = tb.build(synth.DEFAULT_MODEL_CFG)
cfg = synth.DEFAULT_MODEL_CFG
B, N, T, L = 8, 40, 32, 10
gr = {}
for prec in ("f32", "f16x3"):
    m.precision = prec
    inp, loss3, g = tb.train_step_grads(m, cfg, B, N, T, L, 77)
    gr[prec] = {k: v.clone().cpu().double().numpy() for k, v in g.items()}
    print(prec, "loss", loss3.tolist())
torch.set_num_threads(64)
tsd = {k: tb.t64(v, True) for k, v in sd.items()}
sm, st = tb._oracle_forward_grad(tsd, cfg, inp)
neg = tsd["negative_token.weight"].unsqueeze(0).expand(B, -1, -1)
ls = sola_oracle.losses(sm, st, inp["labels"], inp["pos_tokens"], neg, tb.POS_W, tb.TEMP, tb.ALIGN_W, dtype=torch.float64)
ls["total"].backward()
print("oracle loss", float(ls["total"]))
worst = []
for k in gr["f32"]:
    ref = tsd[k].grad.numpy(); mx = np.abs(ref).max() + 1e-30
    worst.append((max(np.abs(gr["f32"][k] - ref).max(), np.abs(gr["f16x3"][k] - ref).max()) / mx, k, np.abs(gr["f32"][k] - ref).max() / mx, np.abs(gr["f16x3"][k] - ref).max() / mx))
for w in sorted(worst, reverse=True)[:12]: print("%-60s f32 err %.2e  f16x3 err %.2e" % (w[1], w[2], w[3]))
# how well conditioned are these gradients?  the same autograd in float32 on the CPU
def t32(x, grad=False): return torch.tensor(np.asarray(x), dtype=torch.float32, requires_grad=grad)
def fwd32(tsd_, cfg, inp):
    obj = t32(inp["object_tokens"]); lang = t32(inp["lang_tokens"])
    x = sola_oracle.encoder(tsd_, cfg, obj)
    pe = sola_oracle.positional_encoding(tsd_, cfg, x.shape[2], torch.float32)
    lang = torch.cat([lang, tsd_["negative_token.weight"].unsqueeze(0).expand(obj.shape[0], -1, -1)], dim=1)
    for layer in range(cfg["n_layers"]): x = sola_oracle.align_layer(tsd_, cfg, layer, x, pe, lang)
    logits = torch.einsum("bntd,bwd->bntw", x, lang).mean(dim=-1)
    a = torch.softmax(logits, dim=-1)
    tok = (x * a.unsqueeze(-1)).sum(dim=2)
    return torch.einsum("bnd,bwd->bnw", tok, lang).mean(dim=-1), tok
tsd32 = {k: t32(v, True) for k, v in sd.items()}
sm, st = fwd32(tsd32, cfg, inp)
neg = tsd32["negative_token.weight"].unsqueeze(0).expand(B, -1, -1)
ls32 = sola_oracle.losses(sm, st, inp["labels"], inp["pos_tokens"], neg, tb.POS_W, tb.TEMP, tb.ALIGN_W, dtype=torch.float32)
ls32["total"].backward()
for k in ["short_motion_encoder.12.weight", "short_motion_encoder.12.bias", "short_motion_encoder.13.bias", "short_motion_encoder.16.weight", "short_motion_encoder.16.bias", "short_motion_encoder.4.weight", "short_motion_encoder.20.weight", "object_lang_align_layers.0.obj_attn.q_proj.weight"]:
    ref = tsd[k].grad.numpy(); mx = np.abs(ref).max()
    print("%-52s cpu-f32 autograd err %.2e | hip f32 %.2e | hip f16x3 %.2e" % (k, np.abs(tsd32[k].grad.double().numpy() - ref).max() / mx, np.abs(gr["f32"][k] - ref).max() / mx, np.abs(gr["f16x3"][k] - ref).max() / mx))
for prec, k in (("f32", "short_motion_encoder.16.weight"), ("f16x3", "short_motion_encoder.12.weight"), ("f32", "short_motion_encoder.16.bias"), ("f16x3", "short_motion_encoder.13.bias")):
    ref = tsd[k].grad.numpy(); e = np.abs(gr[prec][k] - ref) / np.abs(ref).max()
    idx = np.argwhere(e > 0.2 * e.max())
    print(prec, k, "shape", ref.shape, "n bad", len(idx), "worst", np.unravel_index(e.argmax(), e.shape), "bad index ranges per axis", [(int(idx[:, a].min()), int(idx[:, a].max()), len(np.unique(idx[:, a]))) for a in range(idx.shape[1])])
