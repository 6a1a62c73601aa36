"""Round 6: the bf16 training step's 16-bit STORAGE (library precision 3 / module.precision = "bf16", BASELINE config C2 "bf16 training").

q / k / v leave the projection GEMMs as bfloat16 rows, the attention forward and backward read them, and the backward writes dq / dk / dv as
bfloat16 rows - the operand of the dW / dX GEMMs behind it - instead of f32 matrices that a cast pass then read back.  The building blocks are
held to BIT-EXACT statements where one exists:
  * a GEMM that writes bfloat16 writes the round-to-nearest-even of what the same GEMM writes as f32 (bias and residual included);
  * the attention kernels on bfloat16 q / k / v produce, bit for bit, what the f32 kernels produce on the widened values, and their
    bfloat16 outputs are the rounded f32 ones - so any error of the mode is the storage rounding itself, never the kernels;
and the whole step to the tolerances of the bf16-operand step it replaces (tests/test_gpu_backward.py)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sola_amd import _lib, ops, synth  # noqa: E402
from sola_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from sola_amd.loss import track_selection_losses_ragged  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).contiguous()


def attention_f32_high_occupancy(*args, **kw):
    """ops.attention through attn_simple.hip's kernel for every shape (sola_tune attn_variant 2) - the kernel whose bf16 instantiation is
    under test; the default routing sends some of these shapes to kernels with another (equally valid) summation order"""
    check(lib().sola_tune(b"attn_variant", 2), "tune")
    try:
        return ops.attention(*args, **kw)
    finally:
        check(lib().sola_tune(b"attn_variant", 1), "tune")


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 1024), (8192, 512, 768), (300, 256, 128), (2048, 1024, 3072)])
@pytest.mark.parametrize("resid", [False, True])
def test_gemm_bf16_output_is_the_rounded_f32_output(M, N, K, resid):
    torch.manual_seed(M + N + K)
    a, w = bf(torch.randn(M, K, device="cuda")), bf(torch.randn(N, K, device="cuda") / math.sqrt(K))
    bias = torch.randn(N, device="cuda")
    r16 = bf(torch.randn(M, N, device="cuda")) if resid else None
    r32 = r16.float().contiguous() if resid else None
    c32 = torch.empty(M, N, device="cuda")
    c16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    s = current_stream(a.device)
    check(lib().sola_gemm_nt_bf16(ptr(a), K, ptr(w), ptr(bias), ptr(r32), N, 0, ptr(c32), N, 0, M, N, K, s), "gemm f32 out")
    check(lib().sola_gemm_nt_bf16(ptr(a), K, ptr(w), ptr(bias), ptr(r16), N, 1, ptr(c16), N, 1, M, N, K, s), "gemm bf16 out")
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + bias + (r32 if resid else 0)
    assert float((c32 - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))
    assert torch.equal(c16, c32.to(torch.bfloat16))


def _attn_case(name):
    """(G, H, Sq, Sk, inner, q_addr, k_addr, q rows, k rows) of the three attention sites (module/module.py:31-50) at small sizes"""
    H = 8
    if name == "inter_object":  # units = (b, t'), rows n = 0..N-1 at stride T'
        B, N, Tp = 3, 64, 4
        return B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp), B * N * Tp, B * N * Tp
    if name == "inter_object_80":  # more keys than one key group of the four-wave backward holds
        B, N, Tp = 2, 80, 3
        return B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp), B * N * Tp, B * N * Tp
    if name == "motion_24":  # units = tracks, T' = 24 consecutive rows (two-wave backward)
        R, Tp = 40, 24
        return R, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1), R * Tp, R * Tp
    if name == "motion_17x12":  # 17 queries against 12 keys: one-wave backward
        R = 33
        return R, H, 17, 12, 1, (17, 0, 1), (12, 0, 1), R * 17, R * 12
    if name == "motion_4":  # the headline shape's motion attention: T' = 4 (register kernel of <= 4 steps in the backward)
        R, Tp = 130, 4
        return R, H, Tp, Tp, 1, (Tp, 0, 1), (Tp, 0, 1), R * Tp, R * Tp
    if name == "motion_3x2":
        R = 37
        return R, H, 3, 2, 1, (3, 0, 1), (2, 0, 1), R * 3, R * 2
    if name == "object_to_language":  # 512 queries per sample against 48 text ++ negative rows: the chunked backward
        B, M1, W = 3, 512, 48
        return B, H, M1, W, 1, (M1, 0, 1), (W, 0, 1), B * M1, B * W
    raise KeyError(name)


ATTN_CASES = ["inter_object", "inter_object_80", "motion_24", "motion_17x12", "object_to_language"]


@pytest.mark.parametrize("case", ATTN_CASES + ["motion_12", "motion_4"])
def test_attention_forward_on_the_bf16_mfma(case):
    """The shipped bf16 training forward (attn_f16.hip's kernel on v_mfma_f32_16x16x16_bf16, sola_tune "attn_bf16_mfma" 1): q k^T products of
    bfloat16 values are exact in f32, so the scores and the log-sum-exp carry f32 accumulation error only (1e-5); the probabilities enter the
    P V product rounded to bfloat16 (what every 16-bit flash attention does), so the output is within 2^-8 of float64 attention on the widened
    values, relative to the largest |v| a row can see; the bf16 copy is the rounding of the f32 rows."""
    H = 8
    if case == "motion_12":  # one wave per unit: sequences of <= 16 steps
        G, Sq, Sk, inner, qa, ka, qrows, krows = 50, 12, 12, 1, (12, 0, 1), (12, 0, 1), 600, 600
    else:
        G, H, Sq, Sk, inner, qa, ka, qrows, krows = _attn_case(case)
    D = H * 128
    torch.manual_seed(3 + len(case))
    q16, k16, v16 = (bf(torch.randn(n, D, device="cuda") * sc) for n, sc in ((qrows, 1.0), (krows, 1.0), (krows, 1.0)))
    o = torch.zeros(qrows, D, device="cuda")
    o16 = torch.zeros(qrows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(qrows, H, device="cuda")
    scale = 1.0 / math.sqrt(128)
    check(lib().sola_attention_bf16(ptr(q16), D, ptr(k16), D, ptr(v16), D, ptr(o), ptr(o16), D, G, H, 128, Sq, Sk, inner, qa[0], qa[1], qa[2],
                                    ka[0], ka[1], ka[2], scale, ptr(lse), current_stream(q16.device)), "attention_bf16")
    torch.cuda.synchronize()
    assert torch.equal(o16, o.to(torch.bfloat16))
    qd, kd, vd = q16.double().cpu(), k16.double().cpu(), v16.double().cpu()
    worst_o = worst_l = 0.0
    for g in (0, G // 2, G - 1):
        qr = [(g // inner) * qa[0] + (g % inner) * qa[1] + i * qa[2] for i in range(Sq)]
        kr = [(g // inner) * ka[0] + (g % inner) * ka[1] + i * ka[2] for i in range(Sk)]
        for h in (0, H - 1):
            sl = slice(h * 128, (h + 1) * 128)
            sc = qd[qr][:, sl] @ kd[kr][:, sl].t() * scale
            ref = torch.softmax(sc, dim=-1) @ vd[kr][:, sl]
            worst_o = max(worst_o, float((o.cpu().double()[qr][:, sl] - ref).abs().max()) / float(vd[kr][:, sl].abs().max()))
            worst_l = max(worst_l, float((lse.cpu().double()[qr][:, h] - torch.logsumexp(sc, dim=-1)).abs().max()))
    print(f"{case}: bf16-MFMA attention vs float64: output {worst_o:.2e} of max|v|, log-sum-exp {worst_l:.2e}")
    assert worst_o <= 2.0 ** -8 and worst_l <= 2e-5


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_forward_on_bf16_rows_equals_the_f32_kernel_on_the_widened_values(case):
    """(sola_tune "attn_bf16_mfma" 0: the f32-MFMA kernel of attn_simple.hip on bfloat16 rows - the shapes the bf16-MFMA kernel does not take)"""
    check(lib().sola_tune(b"attn_bf16_mfma", 0), "tune")
    try:
        _forward_f32_mfma_case(case)
    finally:
        check(lib().sola_tune(b"attn_bf16_mfma", 1), "tune")


def _forward_f32_mfma_case(case):
    G, H, Sq, Sk, inner, qa, ka, qrows, krows = _attn_case(case)
    D = H * 128
    torch.manual_seed(len(case))
    q16, k16, v16 = (bf(torch.randn(n, D, device="cuda") * sc) for n, sc in ((qrows, 2.0), (krows, 2.0), (krows, 1.0)))
    o_ref, lse_ref = attention_f32_high_occupancy(q16.float(), k16.float(), v16.float(), G, H, Sq, Sk, inner, qa, ka, return_lse=True)
    o = torch.zeros(qrows, D, device="cuda")
    o16 = torch.zeros(qrows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(qrows, H, device="cuda")
    scale = 1.0 / math.sqrt(128)
    check(lib().sola_attention_bf16(ptr(q16), D, ptr(k16), D, ptr(v16), D, ptr(o), ptr(o16), D, G, H, 128, Sq, Sk, inner, qa[0], qa[1], qa[2],
                                    ka[0], ka[1], ka[2], scale, ptr(lse), current_stream(q16.device)), "attention_bf16")
    torch.cuda.synchronize()
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    assert torch.equal(o16, o_ref.to(torch.bfloat16))
    # the bf16 copy alone (no f32 output): the same rows
    o16b = torch.zeros_like(o16)
    check(lib().sola_attention_bf16(ptr(q16), D, ptr(k16), D, ptr(v16), D, None, ptr(o16b), D, G, H, 128, Sq, Sk, inner, qa[0], qa[1], qa[2],
                                    ka[0], ka[1], ka[2], scale, ptr(lse), current_stream(q16.device)), "attention_bf16")
    assert torch.equal(o16b, o16)
    # and it IS attention: float64 softmax(q k^T / sqrt(dh)) v on the widened values
    qd, kd, vd = q16.double().cpu(), k16.double().cpu(), v16.double().cpu()
    g, h = G - 1, H - 1
    qr = [(g // inner) * qa[0] + (g % inner) * qa[1] + i * qa[2] for i in range(Sq)]
    kr = [(g // inner) * ka[0] + (g % inner) * ka[1] + i * ka[2] for i in range(Sk)]
    sl = slice(h * 128, (h + 1) * 128)
    p = torch.softmax(qd[qr][:, sl] @ kd[kr][:, sl].t() * scale, dim=-1)
    assert float((o.cpu().double()[qr][:, sl] - p @ vd[kr][:, sl]).abs().max()) <= 2e-5


@pytest.mark.parametrize("case", ATTN_CASES + ["motion_4", "motion_3x2"])
def test_attention_backward_on_bf16_rows_equals_the_f32_kernel_and_writes_the_rounded_gradients(case):
    """(sola_tune "attn_bwd_bf16_mfma" 0: the f32 products on bfloat16 rows - bit-identical to the f32 kernel on the widened values)"""
    check(lib().sola_tune(b"attn_bwd_bf16_mfma", 0), "tune")
    try:
        _backward_case(case, exact=True)
    finally:
        check(lib().sola_tune(b"attn_bwd_bf16_mfma", 1), "tune")


@pytest.mark.parametrize("case", ATTN_CASES)
def test_attention_backward_on_the_bf16_mfma(case):
    """The shipped bf16 backward (attn_bwd_fused_kernel<.., true, true>: the five products on v_mfma_f32_16x16x16_bf16 with dO, the probabilities
    and dS rounded to bfloat16, f32 accumulation and softmax terms) against the f32 kernel on the same bfloat16 q / k / v: the operand roundings
    are 2^-9 relative per factor, so every gradient row is within 2^-6 of the largest entry of its matrix (measured: ~2^-8)."""
    _backward_case(case, exact=False)


@pytest.mark.parametrize("case", ATTN_CASES + ["motion_4"])
def test_attention_backward_with_a_bf16_output_gradient(case):
    """dO as bfloat16 rows (the out-projection's input-gradient GEMM writes them in the bf16 step, train_bf16_store 3): the kernels widen the
    rows where they widened q - the gradients of the f32-dO launch on the rounded values (the bf16 products round dO to bfloat16 anyway;
    D = dO . O sees the rounded rows in both), and within the bf16-product bound of the f32 kernel."""
    _backward_case(case, exact=False, dout_bf16=True)


@pytest.mark.parametrize("case", ATTN_CASES + ["motion_4"])
def test_attention_backward_with_bf16_output_rows_too(case):
    """... and O as bfloat16 rows (the forward of the bf16 step keeps only the rows the out-projection reads): O enters D = dO . O only; the
    launch equals the f32-O launch on the rounded rows (as above), and stays within the bf16-product bound of the f32 kernel on the f32 O."""
    _backward_case(case, exact=False, dout_bf16=True, o_bf16=True)


def _backward_case(case, exact, dout_bf16=False, o_bf16=False):
    G, H, Sq, Sk, inner, qa, ka, qrows, krows = _attn_case(case)
    D = H * 128
    torch.manual_seed(100 + len(case))
    q16, k16, v16 = (bf(torch.randn(n, D, device="cuda") * sc) for n, sc in ((qrows, 1.5), (krows, 1.5), (krows, 1.0)))
    qf, kf, vf = q16.float(), k16.float(), v16.float()
    o, lse = ops.attention(qf, kf, vf, G, H, Sq, Sk, inner, qa, ka, return_lse=True)
    dout = torch.randn(qrows, D, device="cuda") * 1e-3
    dout16 = bf(dout)
    if dout_bf16:
        dout = dout16.float()  # the reference launches see the rounded rows
    dq, dk, dv = ops.attention_backward(qf, kf, vf, o, dout, lse, G, H, Sq, Sk, inner, qa, ka)
    # gradients into column slices of wider matrices, as backward.hip lays them out ([rows][3D])
    g16 = torch.full((qrows, 3 * D), 7.0, device="cuda", dtype=torch.bfloat16)
    gk16 = g16 if krows == qrows else torch.full((krows, 3 * D), 7.0, device="cuda", dtype=torch.bfloat16)
    dq_scr = torch.empty(qrows, 3 * D, device="cuda")
    dvec = torch.empty(qrows, H, device="cuda")
    n_scr = int(lib().sola_attention_backward_scratch_floats(qrows, G, H, Sk))
    scr = torch.empty(max(n_scr, 1), device="cuda")
    e2 = 2  # bytes per value
    dout_arg = dout16 if dout_bf16 else dout
    o_arg = bf(o) if o_bf16 else o
    check(lib().sola_attention_backward_bf16(ptr(q16), D, ptr(k16), D, ptr(v16), D, C.c_void_p(o_arg.data_ptr()), 1 if o_bf16 else 0,
                                             C.c_void_p(dout_arg.data_ptr()), 1 if dout_bf16 else 0, D, ptr(lse),
                                             C.c_void_p(g16.data_ptr()), C.c_void_p(gk16.data_ptr() + D * e2), C.c_void_p(gk16.data_ptr() + 2 * D * e2),
                                             3 * D, 3 * D, 3 * D, ptr(dq_scr), ptr(dvec), G, H, 128, Sq, Sk, inner, qa[0], qa[1], qa[2], ka[0], ka[1], ka[2],
                                             1.0 / math.sqrt(128), qrows, ptr(scr) if n_scr else None, n_scr, current_stream(q16.device)), "attention_backward_bf16")
    torch.cuda.synchronize()
    if dout_bf16:  # the same launch with the (rounded) rows handed over as f32: identical bits
        h16 = torch.full_like(g16, 7.0)
        hk16 = h16 if krows == qrows else torch.full_like(gk16, 7.0)
        o_ref = bf(o).float() if o_bf16 else o  # (a bf16 O: the reference launch sees the rounded rows)
        check(lib().sola_attention_backward_bf16(ptr(q16), D, ptr(k16), D, ptr(v16), D, ptr(o_ref), 0, C.c_void_p(dout.data_ptr()), 0, D, ptr(lse),
                                                 C.c_void_p(h16.data_ptr()), C.c_void_p(hk16.data_ptr() + D * e2), C.c_void_p(hk16.data_ptr() + 2 * D * e2),
                                                 3 * D, 3 * D, 3 * D, ptr(dq_scr), ptr(dvec), G, H, 128, Sq, Sk, inner, qa[0], qa[1], qa[2], ka[0], ka[1], ka[2],
                                                 1.0 / math.sqrt(128), qrows, ptr(scr) if n_scr else None, n_scr, current_stream(q16.device)), "attention_backward_bf16")
        torch.cuda.synchronize()
        # the four-wave shape sums D = dO . O in another order on the raw rows (sixteen lanes a row), the register kernel may contract its
        # multiply-adds differently: f32 differences of an ulp, visible only where a gradient sits on a bfloat16 rounding boundary
        for x, y in ((h16, g16), (hk16, gk16)):
            assert float((x.float() - y.float()).abs().max()) <= 2.0 ** -7 * float(y.float().abs().max())
            assert float((x != y).float().mean()) <= 5e-3
        if case in ("motion_24", "motion_17x12"):  # register-staged shapes: the same arithmetic in the same order
            assert torch.equal(h16, g16) and torch.equal(hk16, gk16)
    if exact and max(Sq, Sk) <= 4:
        # the register kernel of <= 4 steps is plain f32 VALU code: its two instantiations may contract their multiply-adds differently, so
        # the f32 gradients agree to an ulp and their bfloat16 roundings to one bfloat16 ulp (2^-8 relative)
        for got, ref in ((g16[:, :D], dq), (gk16[:, D:2 * D], dk), (gk16[:, 2 * D:], dv)):
            r16 = ref.to(torch.bfloat16).float()
            assert float(((got.float() - r16).abs() - 2.0 ** -7 * r16.abs()).max()) <= 1e-12
            assert float((got.float() != r16).float().mean()) <= 1e-3  # ... and only where the f32 value sits on a rounding boundary
    elif exact:
        assert torch.equal(g16[:, :D], dq.to(torch.bfloat16))
        assert torch.equal(gk16[:, D:2 * D], dk.to(torch.bfloat16))
        assert torch.equal(gk16[:, 2 * D:], dv.to(torch.bfloat16))
    else:
        for name, got, ref in (("dq", g16[:, :D], dq), ("dk", gk16[:, D:2 * D], dk), ("dv", gk16[:, 2 * D:], dv)):
            err = float((got.float() - ref).abs().max()) / float(ref.abs().max())
            cos = float((got.float() * ref).sum() / (got.float().norm() * ref.norm()))
            print(f"{case} {name}: max err / max|ref| {err:.2e}, cosine {cos:.6f}")
            assert err <= 2.0 ** -6 and cos >= 0.9995, (name, err, cos)
    if krows != qrows:
        assert torch.all(g16[:, D:] == 7.0)  # nothing outside the addressed slices


def _ragged_step(m, smp, seed):
    objs, langs = [x["obj"] for x in smp], [x["lang"] for x in smp]
    labels = torch.cat([x["labels"] for x in smp])
    pos = torch.stack([x["pos"] for x in smp])
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(seed)
    m.forward_ragged(objs, langs, differentiable=True)
    flat, tok, offs, counts = m.last_ragged
    loss = track_selection_losses_ragged(flat, tok, labels, pos, m.negative_token.weight, offs, counts, 1.5, 0.07, 0.3)
    loss[:, 0].mean().backward()
    torch.cuda.synchronize()
    return loss.detach().cpu().double(), {k: p.grad.detach().double().clone() for k, p in m.named_parameters()}, flat.detach().cpu()


def _cos(a, b):
    num = sum(float((a[k] * b[k]).sum()) for k in a)
    return num / math.sqrt(sum(float(a[k].pow(2).sum()) for k in a) * sum(float(b[k].pow(2).sum()) for k in a))


@pytest.mark.parametrize("variant", ["base", "lin_div64"])
def test_ragged_bf16_step_with_bf16_q_k_v_storage(variant):
    """One ragged optimizer-step gradient (24 samples of the MeViS-like mix, dropout on, same mask seed) in three arithmetics: exact f32, bf16
    GEMM operands with f32 q / k / v / dq / dk / dv (sola_tune train_bf16_store 0: the round-5 step) and with them stored as bfloat16 (the
    default now).  Every attention site of the ragged step takes the bf16 rows (the library reports which did).  The storage mode must sit in
    the operand mode's error class against exact f32: cosine of the whole gradient within 0.02 of it at random-init weights (saturated first
    softmax; measured 0.591 operands / 0.586 stored, per-sample losses within 2.6 % / 2.7 %) and within 0.004 on weights with an unsaturated
    softmax (0.99583 both, losses within 0.42 % both); the per-sample losses no further off than 1.25 x the operand mode's + 0.1 %."""
    cfg = synth.DEFAULT_MODEL_CFG
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict_variant(cfg, 42, variant).items()}, strict=True)
    m = m.cuda().train()
    smp = synth.make_ragged_samples(cfg, 24, 2024, "cuda")
    res = {}
    for tag, prec, store in (("f32", "f32", 3), ("operands", "bf16", 0), ("stored", "bf16", 3)):
        m.precision = prec
        check(lib().sola_tune(b"train_bf16_store", store), "tune")
        try:
            if tag == "stored":  # the bfloat16 pre-norm rows (level 2) read their residual from the kept-operand arena, sized from the
                _ragged_step(m, smp, 98)  # previous step's need: the second step of a run is the first with the whole storage mode on
            res[tag] = _ragged_step(m, smp, 99)
        finally:
            check(lib().sola_tune(b"train_bf16_store", 3), "tune")
    m.precision = "f32"
    l32, g32, _ = res["f32"]
    out = {}
    for tag in ("operands", "stored"):
        l, g, _ = res[tag]
        out[tag] = (float((l[:, 0] / l32[:, 0] - 1).abs().max()), _cos(g, g32))
    print(f"{variant}: loss rel err / gradient cosine vs exact f32 - bf16 operands {out['operands']}, bf16 q/k/v storage {out['stored']}; "
          f"stored vs operands cosine {_cos(res['stored'][1], res['operands'][1]):.4f}")
    assert not torch.equal(res["stored"][2], res["operands"][2])  # the storage mode really ran (q / k / v rounded: the logits move)
    assert out["stored"][0] <= 1.25 * out["operands"][0] + 1e-3, out
    slack = 0.02 if variant == "base" else 0.004
    assert out["stored"][1] >= out["operands"][1] - slack, out


def _oracle_step(sd, cfg, inp, B, autocast):
    """one training-step gradient through the CPU oracle (checker): fp32 (autocast None), or under torch.autocast(<16-bit dtype>) - the
    textbook mixed-precision recipe"""
    from oracle import sola_oracle  # checker only

    tsd = {k: torch.tensor(v, requires_grad=(k != "positional_encoding_gaussian_matrix")) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=autocast or torch.bfloat16, enabled=autocast is not None):
        sm, st = sola_oracle.forward(tsd, cfg, inp["object_tokens"], inp["lang_tokens"])
    neg = tsd["negative_token.weight"].unsqueeze(0).repeat(B, 1, 1)
    ls = sola_oracle.losses(sm.float(), st.float(), inp["labels"], inp["pos_tokens"], neg, 1.5, 0.07, 0.3)
    ls["total"].backward()
    return float(ls["total"]), {k: v.grad.double() for k, v in tsd.items() if v.grad is not None}


@pytest.mark.parametrize("variant,mode", [("base", "bf16"), ("lin_div64", "bf16"), ("base", "f16")])  # (lin_div64, f16) measured 0.9995 / 0.9994 / 0.9995; a minute of CPU half-precision matmuls
def test_16_bit_step_against_the_oracle_under_torch_autocast(variant, mode):
    """the bf16 step (bfloat16 GEMM operands, q / k / v and the gradient operands stored as bfloat16) against the ORACLE, not
    against the library's own f32 step: autograd through oracle/sola_oracle.py in fp32 and under torch.autocast("cpu", bfloat16), 8 samples
    of (N=40, T=32, L=10), dropout off.  The library's step must be at least as close to the fp32 oracle as autocast is (loss and gradient
    cosine; its activations, statistics, softmax and residuals stay f32, autocast's GroupNorm inputs and attention run in bf16), and on
    weights whose first softmax is not saturated ("lin_div64"), where two bf16 evaluations are comparable at all, it must agree with the
    autocast gradient itself (cosine >= 0.985).  The same for the f16-operand step (module.precision = "f16": f32 storage, per-tensor power-of-two
    scales) against torch.autocast(float16): measured 0.985 library / 0.986 autocast at random init.  bf16, measured: base - library 0.742 / autocast 0.550 against fp32 (library vs autocast 0.64:
    two noisy evaluations of a saturated softmax), losses 8.497 / 8.480 against 8.464; lin_div64 - 0.9952 / 0.9947, library vs autocast 0.9954."""
    from sola_amd.loss import track_selection_losses

    cfg = synth.DEFAULT_MODEL_CFG
    B, N, T, L = 8, 40, 32, 10
    sd = synth.make_state_dict_variant(cfg, 42, variant)
    inp = synth.make_inputs(cfg, B, N, T, L, 77)
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
    l32, g32 = _oracle_step(sd, cfg, inp, B, None)
    l16, g16 = _oracle_step(sd, cfg, inp, B, torch.bfloat16 if mode == "bf16" else torch.float16)
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.precision = mode
    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    for _ in range(2):  # the second step of a run has the kept-operand arena, i.e. the whole storage mode (bfloat16 pre-norm rows too)
        m.zero_grad(set_to_none=True)
        sm, st = m(c["object_tokens"], c["lang_tokens"])
        neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
        loss3 = track_selection_losses(sm, st, c["labels"], c["pos_tokens"], neg, 1.5, 0.07, 0.3)
        loss3[0].backward()
        torch.cuda.synchronize()
    gh = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
    assert set(gh) == set(g32)
    lh = float(loss3[0])
    c_hip, c_auto, c_both = _cos(gh, g32), _cos(g16, g32), _cos(gh, g16)
    print(f"{variant} {mode}: loss fp32 oracle {l32:.5f}, autocast oracle {l16:.5f}, library {lh:.5f}; gradient cosine vs fp32 oracle: library {c_hip:.4f}, "
          f"autocast {c_auto:.4f}; library vs autocast {c_both:.4f}")
    assert abs(lh / l32 - 1) <= max(2 * abs(l16 / l32 - 1), 5e-3)
    assert c_hip >= c_auto - 0.02
    if variant == "lin_div64":
        assert c_hip >= 0.99 and c_both >= 0.985
