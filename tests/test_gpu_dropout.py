"""Train-mode dropout (nn.Dropout p=0.2 after every encoder LeakyReLU, module/module.py:78-94; SDPA dropout p=0.1 on
the attention probabilities, tools/attention.py:12,71).  torch's RNG stream cannot be reproduced bit-wise, so parity
is defined as: (1) the masks have the right rate and scaling, (2) forward and backward use the SAME mask - checked
exactly by recovering the mask from the forward output and differentiating the oracle with it, (3) the full training
step is deterministic per seed, changes with the seed, and equals the eval numerics when p = 0."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sola_oracle  # noqa: E402
from sola_amd import ops, synth  # noqa: E402
from sola_amd.loss import track_selection_losses  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402


def cuda(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def t64(x, grad=False):
    return torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=grad)


@pytest.fixture(autouse=True)
def _reset_stage_dropout():
    yield
    ops.set_stage_dropout(0.0, 0)


def test_encoder_dropout_mask_rate_scale_and_backward():
    rng = np.random.default_rng(0)
    R, Tl, C, p = 64, 16, 512, 0.2
    x = (rng.standard_normal((R * Tl, C)) * 2 + 0.3).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    dy = rng.standard_normal((R * Tl, C)).astype(np.float32)
    y0 = ops.group_norm(cuda(x), cuda(gamma), cuda(beta), 8, R, 1, Tl, 0, 1, Tl, leaky_slope=0.01).cpu().numpy()
    ops.set_stage_dropout(p, 1234)
    y1 = ops.group_norm(cuda(x), cuda(gamma), cuda(beta), 8, R, 1, Tl, 0, 1, Tl, leaky_slope=0.01).cpu().numpy()
    keep = y1 != 0
    assert abs(keep.mean() - (1 - p)) < 3e-3  # 524k Bernoulli draws: sigma = 5.5e-4
    np.testing.assert_allclose(y1[keep], y0[keep] / (1 - p), rtol=1e-6, atol=1e-7)
    assert abs(keep[:, :256].mean() - keep[:, 256:].mean()) < 5e-3 and abs(keep[::2].mean() - keep[1::2].mean()) < 5e-3
    # backward with the same (seed, index) mask == autograd of the oracle multiplied by the recovered mask
    xt, gt, bt = t64(x, True), t64(gamma, True), t64(beta, True)
    yr = sola_oracle.leaky_relu(sola_oracle.group_norm_tokens(xt.reshape(R, Tl, C), gt, bt, 8)).reshape(R * Tl, C)
    (yr * t64(keep.astype(np.float64)) / (1 - p)).backward(t64(dy))
    dx, dg, db = ops.group_norm_backward(cuda(x), cuda(dy), cuda(gamma), cuda(beta), 8, R, 1, Tl, 0, 1, Tl, leaky_slope=0.01)
    for got, ref, nm in ((dx, xt.grad, "dx"), (dg, gt.grad, "dgamma"), (db, bt.grad, "dbeta")):
        err = float(np.abs(got.cpu().double().numpy() - ref.numpy()).max())
        assert err <= 3e-5 * float(ref.abs().max()), (nm, err)
    # another seed gives another mask
    ops.set_stage_dropout(p, 99)
    y2 = ops.group_norm(cuda(x), cuda(gamma), cuda(beta), 8, R, 1, Tl, 0, 1, Tl, leaky_slope=0.01).cpu().numpy()
    assert ((y2 != 0) != keep).mean() > 0.2


@pytest.mark.parametrize("layout", ["obj", "motion", "motion_packed", "o2l"])
def test_attention_dropout_same_mask_forward_and_backward(layout):
    """V = one-hot rows exposes the dropped probability matrix as the attention output, which yields the mask."""
    H, dh, p = 8, 16, 0.1
    D = H * dh
    rng = np.random.default_rng(3)
    if layout == "obj":
        B, N, Tp = 2, 13, 3
        G, Sq, Sk, inner, qa, ka = B * Tp, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp)
        view = lambda t: t.reshape(B, N, Tp, D).permute(0, 2, 1, 3).reshape(G, N, D)
        rows_q = rows_k = B * N * Tp
    elif layout in ("motion", "motion_packed"):
        B, N, Tp = (2, 5, 9) if layout == "motion" else (3, 7, 4)  # T'=4: four (track, head) units share one MFMA tile
        G, Sq, Sk, inner, qa, ka = B * N, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1)
        view = lambda t: t.reshape(G, Tp, D)
        rows_q = rows_k = B * N * Tp
    else:
        B, Sq, Sk = 2, 40, 15
        G, inner, qa, ka = B, 1, (Sq, 0, 1), (Sk, 0, 1)
        view = None
        rows_q, rows_k = B * Sq, B * Sk
    q = rng.standard_normal((rows_q, D)).astype(np.float32)
    k = rng.standard_normal((rows_k, D)).astype(np.float32)
    v = rng.standard_normal((rows_k, D)).astype(np.float32)
    do = rng.standard_normal((rows_q, D)).astype(np.float32)
    vq = (lambda t: view(t)) if view else (lambda t: t.reshape(B, Sq, D))
    vk = (lambda t: view(t)) if view else (lambda t: t.reshape(B, Sk, D))
    # probe: V[key, h*dh + j] = [j == key index within its group]  (Sk <= dh)
    key_idx = np.zeros(rows_k, dtype=np.int64)
    kk = vk(torch.arange(rows_k).reshape(rows_k, 1).expand(rows_k, D).double())[:, :, 0].long()  # [G, Sk] row ids
    for j in range(Sk):
        key_idx[kk[:, j].numpy()] = j
    probe = np.zeros((rows_k, H, dh), dtype=np.float32)
    probe[np.arange(rows_k), :, key_idx] = 1.0
    probe = probe.reshape(rows_k, D)
    ops.set_stage_dropout(p, 777)
    pd = ops.attention(cuda(q), cuda(k), cuda(probe), G, H, Sq, Sk, inner, qa, ka)  # = dropped P, per head
    pdv = vq(pd.cpu().double()).reshape(G, Sq, H, dh)[..., :Sk].permute(0, 2, 1, 3)  # [G,H,Sq,Sk]
    mask = (pdv != 0).double()
    assert abs(mask.mean().item() - (1 - p)) < 0.03
    # oracle with the recovered mask
    qt, kt, vt = t64(q, True), t64(k, True), t64(v, True)
    qh = vq(qt).reshape(G, Sq, H, dh).permute(0, 2, 1, 3)
    kh = vk(kt).reshape(G, Sk, H, dh).permute(0, 2, 1, 3)
    vh = vk(vt).reshape(G, Sk, H, dh).permute(0, 2, 1, 3)
    P = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    np.testing.assert_allclose(pdv.numpy(), (P.detach() * mask / (1 - p)).numpy(), atol=2e-6)  # forward parity incl. scale
    o_ref = ((P * mask / (1 - p)) @ vh).permute(0, 2, 1, 3).reshape(G, Sq, D)
    o_ref.backward(vq(t64(do)))
    o, lse = ops.attention(cuda(q), cuda(k), cuda(v), G, H, Sq, Sk, inner, qa, ka, return_lse=True)
    np.testing.assert_allclose(vq(o.cpu().double()).numpy(), o_ref.detach().numpy(), atol=3e-6)
    dq, dk, dv = ops.attention_backward(cuda(q), cuda(k), cuda(v), o, cuda(do), lse, G, H, Sq, Sk, inner, qa, ka)
    for got, ref, nm in ((dq, qt.grad, "dq"), (dk, kt.grad, "dk"), (dv, vt.grad, "dv")):
        err = float(np.abs(got.cpu().double().numpy() - ref.numpy()).max())
        assert err <= 3e-5 * float(ref.abs().max()), (layout, nm, err)


def _train_step(m, cfg, B, N, T, L, seed_inputs):
    inp = synth.make_inputs(cfg, B, N, T, L, seed_inputs)
    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    m.zero_grad(set_to_none=True)
    sm, st = m(c["object_tokens"], c["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
    loss3 = track_selection_losses(sm, st, c["labels"], c["pos_tokens"], neg, 1.5, 0.07, 0.3)
    loss3[0].backward()
    return sm.detach().clone(), loss3.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}


def test_training_step_with_dropout_is_seeded_and_consistent():
    cfg = synth.SMALL_MODEL_CFG  # dropout_p = 0.2, attention dropout 0.1: the reference's train() behaviour
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
    m = m.cuda()
    shape = (2, 6, 24, 7)
    m.eval()
    sm_eval, loss_eval, g_eval = _train_step(m, cfg, *shape, 5)
    m.train()
    torch.manual_seed(11)
    sm_a, loss_a, g_a = _train_step(m, cfg, *shape, 5)
    torch.manual_seed(11)
    sm_b, loss_b, g_b = _train_step(m, cfg, *shape, 5)
    torch.manual_seed(12)
    sm_c, _, _ = _train_step(m, cfg, *shape, 5)
    assert torch.equal(sm_a, sm_b) and torch.equal(loss_a, loss_b) and all(torch.equal(g_a[k], g_b[k]) for k in g_a)
    assert (sm_a - sm_c).abs().max() > 1e-3 and (sm_a - sm_eval).abs().max() > 1e-3
    assert all(torch.isfinite(g).all() for g in g_a.values())
    # directional derivative of the seeded (hence deterministic, smooth) loss agrees with the HIP gradient
    params = dict(m.named_parameters())
    gen = torch.Generator(device="cuda").manual_seed(0)
    direction = {k: torch.randn(p.shape, device="cuda", generator=gen) * (p.detach().abs().mean() + 1e-3) for k, p in params.items()}
    analytic = sum(float((g_a[k].double() * direction[k].double()).sum()) for k in params)
    eps = 1e-4  # fp32 finite differences of this loss are good to a few % (same spread in eval mode, tools/dd_probe.py)
    vals = []
    saved = {k: p.detach().clone() for k, p in params.items()}
    for sgn in (+1, -1):
        with torch.no_grad():
            for k, p in params.items():
                p.add_(sgn * eps * direction[k])
        torch.manual_seed(11)
        _, l, _ = _train_step(m, cfg, *shape, 5)
        vals.append(float(l[0]))
        with torch.no_grad():
            for k, p in params.items():
                p.copy_(saved[k])  # add-then-subtract does not restore fp32 weights bit-exactly
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(numeric - analytic) <= 0.08 * abs(analytic) + 1e-3, (numeric, analytic)  # a wrong mask is O(1) off
    # p = 0 in train mode reproduces the eval numerics exactly
    m.dropout_p = 0.0
    m.attention_dropout_p = 0.0
    sm_0, loss_0, g_0 = _train_step(m, cfg, *shape, 5)
    assert torch.equal(sm_0, sm_eval) and torch.equal(loss_0, loss_eval)
