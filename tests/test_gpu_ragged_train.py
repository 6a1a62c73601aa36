"""Ragged TRAINING step (sola_forward_train_ragged / sola_backward_ragged / sola_loss_backward_ragged): one optimizer step over
samples of DIFFERENT (N, T, L), the shapes real data has (dataloader.py:119-163; the reference trains one sample per step,
configs/mevis/default.yaml:37, train.py:62-137).

* the summed gradient of a mixed-shape batch equals the SUM of the reference's own per-sample gradients (tests/golden, full
  tensors) and the sum of this library's one-sample training steps, in every precision mode;
* per-sample losses equal the reference's; the mean of the per-sample totals gives the averaged gradient;
* shapes: T' = 1, odd lengths, one track, more than 64 tracks, more than 16 encoded steps, object->language units of fewer
  than 16 queries, text lengths 1..40;
* dropout: an equal-shape ragged batch reproduces the uniform batch under the same mask seed (forward and gradients);
* call-sequence errors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import _load  # noqa: E402

from conftest import case_dict  # noqa: E402,F401
from sola_amd import SolaError, _lib, synth  # noqa: E402
from sola_amd.loss import track_selection_losses, track_selection_losses_ragged  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

POS_W, TEMP, ALIGN_W = 1.5, 0.07, 0.3


def build(cfg, precision="f32"):
    m = LanguageAlignedTrackSelectionModule(cfg)
    sd = synth.make_state_dict(cfg, 42)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()  # eval(): dropout off, as in the golden run; gradients still flow
    m.precision = precision
    return m


@pytest.fixture(scope="module")
def small():
    return build(synth.SMALL_MODEL_CFG)


@pytest.fixture(scope="module")
def full():
    return build(synth.DEFAULT_MODEL_CFG)


@pytest.fixture(autouse=True)
def no_size_gate():
    """the test batches are below the production size gate of the reduced-precision training GEMMs"""
    _lib.check(_lib.lib().sola_tune(b"train_split_min_rows", 0), "tune")
    yield
    _lib.check(_lib.lib().sola_tune(b"train_split_min_rows", 1024), "tune")


def sample_inputs(cfg, N, T, L, seed):
    inp = synth.make_inputs(cfg, 1, N, T, L, seed)
    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    return {"obj": c["object_tokens"][0], "lang": c["lang_tokens"][0], "labels": c["labels"][0], "pos": c["pos_tokens"][0, 0]}


def one_sample_step(m, smp):
    """the reference's step: batch size 1, mean over the sample's tracks (train.py:62-117)"""
    m.zero_grad(set_to_none=True)
    sm, st = m(smp["obj"][None], smp["lang"][None])
    neg = m.negative_token.weight.clone().unsqueeze(0)
    loss3 = track_selection_losses(sm, st, smp["labels"][None], smp["pos"][None, None], neg, POS_W, TEMP, ALIGN_W)
    loss3[0].backward()
    return loss3.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, sm.detach()[0].clone()


def ragged_step(m, samples, weights=None):
    """one step over all samples: loss = sum_i w_i * total_i (w_i = 1: the sum of the per-sample objectives)"""
    m.zero_grad(set_to_none=True)
    sms, sts = m.forward_ragged([s["obj"] for s in samples], [s["lang"] for s in samples], differentiable=True)
    flat, tok, offs, counts = m.last_ragged
    assert flat.requires_grad and tok.requires_grad
    labels = torch.cat([s["labels"] for s in samples])
    pos = torch.stack([s["pos"] for s in samples])
    loss = track_selection_losses_ragged(flat, tok, labels, pos, m.negative_token.weight, offs, counts, POS_W, TEMP, ALIGN_W)
    w = torch.ones(len(samples), device=flat.device) if weights is None else weights
    (loss[:, 0] * w).sum().backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, [t.detach().clone() for t in sms]


def assert_grads_close(got, ref, rel, what):
    total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    bad = {}
    for k, r in ref.items():
        err = float((got[k] - r).abs().max())
        tol = rel * float(r.abs().max()) + 1e-6 * total
        if err > tol:
            bad[k] = (err, float(r.abs().max()))
    assert not bad, f"{what}: gradient mismatch {bad}"


@pytest.fixture(scope="module")
def rt_golden():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "ragged_train_golden.npz"), allow_pickle=False)


def golden_samples(rt_golden, tag, cfg):
    shapes = [tuple(int(v) for v in row) for row in rt_golden[f"{tag}.shapes"]]
    seed0 = int(rt_golden[f"{tag}.seed0"])
    return shapes, [sample_inputs(cfg, N, T, L, seed0 + i) for i, (N, T, L) in enumerate(shapes)]


def test_mixed_batch_equals_the_sum_of_the_reference_gradients_small(rt_golden, small):
    """tests/golden/ragged_train_golden.npz (gen_golden.py ragged_train): the REFERENCE stepped one sample at a time over ten
    samples of different (N, T, L), gradients accumulated.  One ragged step with loss = sum of the per-sample totals must give
    that summed gradient (all 83 tensors in full), the reference's per-sample losses and its logits."""
    cfg = synth.SMALL_MODEL_CFG
    shapes, samples = golden_samples(rt_golden, "small", cfg)
    loss, got, sms = ragged_step(small, samples)
    np.testing.assert_allclose(loss.cpu().numpy().astype(np.float64), rt_golden["small.loss"], rtol=2e-4, atol=2e-4)
    for i, sm in enumerate(sms):
        np.testing.assert_allclose(sm.cpu().numpy(), rt_golden[f"small.s{i}.score_map"], rtol=0, atol=1e-3, err_msg=f"sample {i} {shapes[i]}")
    ref = {k[len("small.gradsum."):]: torch.from_numpy(rt_golden[k]).cuda() for k in rt_golden.files if k.startswith("small.gradsum.")}
    assert set(ref) == set(got) and len(ref) == 83
    assert_grads_close(got, ref, 2e-3, "vs the reference's accumulated gradients")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_mixed_batch_equals_the_sum_of_the_reference_gradients_full(rt_golden, full, precision):
    """Default configuration (33 M parameters): per-parameter norms of the reference's accumulated gradient and the first 256
    elements of every tensor; exact-f32 and split-f16 GEMMs at the same bar as the one-sample gradient tests."""
    cfg = synth.DEFAULT_MODEL_CFG
    shapes, samples = golden_samples(rt_golden, "full", cfg)
    full.precision = precision
    try:
        loss, got, sms = ragged_step(full, samples)
    finally:
        full.precision = "f32"
    np.testing.assert_allclose(loss.cpu().numpy().astype(np.float64), rt_golden["full.loss"], rtol=3e-4, atol=3e-4)
    for i, sm in enumerate(sms):
        np.testing.assert_allclose(sm.cpu().numpy(), rt_golden[f"full.s{i}.score_map"], rtol=0, atol=1e-3, err_msg=f"sample {i} {shapes[i]}")
    total = float(np.sqrt(sum(float(rt_golden[k]) ** 2 for k in rt_golden.files if k.startswith("full.gradsum_norm."))))
    bad = {}
    for k, g in got.items():
        ref_n = float(rt_golden["full.gradsum_norm." + k])
        got_n = float(g.double().norm())
        if abs(got_n - ref_n) > 2e-3 * ref_n + 1e-5 * total:
            bad[k] = ("norm", got_n, ref_n)
        head = rt_golden["full.gradsum_head." + k]
        err = float(np.abs(g.reshape(-1)[:head.size].cpu().numpy() - head).max())
        if err > 3e-3 * float(g.abs().max()) + 1e-6 * total:
            bad[k] = ("head", err, float(g.abs().max()))
    assert not bad, f"{precision}: {bad}"


SHAPES = [(4, 24, 2), (33, 9, 6), (16, 32, 9), (1, 1, 1), (9, 57, 16), (66, 140, 30), (2, 8, 3)]


def test_mixed_batch_equals_one_sample_steps_small(small):
    cfg = synth.SMALL_MODEL_CFG
    samples = [sample_inputs(cfg, N, T, L, 500 + i) for i, (N, T, L) in enumerate(SHAPES)]
    ref_sum, ref_loss, ref_sm = None, [], []
    for s in samples:
        l3, g, sm = one_sample_step(small, s)
        ref_loss.append(l3)
        ref_sm.append(sm)
        ref_sum = g if ref_sum is None else {k: ref_sum[k] + g[k] for k in g}
    loss, got, sms = ragged_step(small, samples)
    torch.testing.assert_close(loss, torch.stack(ref_loss), rtol=2e-4, atol=2e-4)
    for a, b in zip(sms, ref_sm):
        torch.testing.assert_close(a, b, rtol=0, atol=2e-4)
    assert_grads_close(got, ref_sum, 2e-3, "vs one-sample steps")
    # the mean of the per-sample totals (what train.py steps on) = the average of the per-sample gradients
    w = torch.full((len(samples),), 1.0 / len(samples), device="cuda")
    _, avg, _ = ragged_step(small, samples, w)
    assert_grads_close(avg, {k: v / len(samples) for k, v in ref_sum.items()}, 2e-3, "mean objective")
    # bit-reproducible (fixed reduction orders everywhere)
    _, again, _ = ragged_step(small, samples, w)
    for k in avg:
        assert torch.equal(avg[k], again[k]), k


FULL_SHAPES = [(8, 8, 16), (64, 32, 16), (16, 32, 9), (3, 1, 4), (80, 32, 11), (7, 33, 24), (12, 150, 5)]


@pytest.mark.parametrize("precision", ["f32", "f16x3", "f16", "bf16"])
def test_mixed_batch_equals_one_sample_steps_full(full, precision):
    """Default configuration, all four precision modes: each mode's ragged step against its own one-sample steps (which
    test_gpu_backward.py holds to the reference's gradients)."""
    cfg = synth.DEFAULT_MODEL_CFG
    full.precision = precision
    try:
        samples = [sample_inputs(cfg, N, T, L, 400 + i) for i, (N, T, L) in enumerate(FULL_SHAPES)]
        ref_sum, ref_loss = None, []
        for s in samples:
            l3, g, _ = one_sample_step(full, s)
            ref_loss.append(l3)
            ref_sum = g if ref_sum is None else {k: ref_sum[k] + g[k] for k in g}
        loss, got, _ = ragged_step(full, samples)
    finally:
        full.precision = "f32"
    if precision in ("f16", "bf16"):  # reduced precision with a stated tolerance (tests/test_gpu_backward.py): per-tensor scales differ
        # between a one-sample call and the batch, so the comparison is statistical
        tol, min_cos = (2e-2, 0.95) if precision == "f16" else (5e-2, 0.95)  # measured: 2.5e-3 / 0.99993 and 1.4e-2 / 0.99972
        num = sum(float((got[k].double() * ref_sum[k].double()).sum()) for k in got)
        den = (sum(float((got[k].double() ** 2).sum()) for k in got) * sum(float((ref_sum[k].double() ** 2).sum()) for k in got)) ** 0.5
        print(f"{precision}: ragged step vs one-sample steps, worst loss difference {float((loss - torch.stack(ref_loss)).abs().max()):.3e}, "
              f"gradient cosine {num / den:.5f}")
        torch.testing.assert_close(loss, torch.stack(ref_loss), rtol=tol, atol=tol)
        assert num / den > min_cos, f"gradient cosine {num / den}"
        return
    torch.testing.assert_close(loss, torch.stack(ref_loss), rtol=3e-4, atol=3e-4)
    assert_grads_close(got, ref_sum, 5e-3 if precision == "f16x3" else 2e-3, f"vs one-sample steps ({precision})")


@pytest.mark.parametrize("precision", ["f16x3", "f16", "bf16"])
def test_kept_operand_casts_give_the_same_ragged_gradients_bit_for_bit(full, precision):
    """As tests/test_gpu_backward.py's test of the same name, on a mixed-shape ragged batch (conv taps through the row maps): the
    backward reading the forward's kept operand casts (sola_tune "train_x16_keep" 1) against casting again (0).  (sola_tune
    "train_bf16_store" 1: level 2's bfloat16 pre-norm rows exist only with the arena - a capped arena keeps some residual copies and not
    others - so this A/B of the arena itself runs without them.)"""
    cfg = synth.DEFAULT_MODEL_CFG
    full.precision = precision
    got = {}
    try:
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 1), "tune")
        samples = [sample_inputs(cfg, N, T, L, 400 + i) for i, (N, T, L) in enumerate(FULL_SHAPES)]
        for keep in (1, 0):
            _lib.check(_lib.lib().sola_tune(b"train_x16_keep", keep), "tune")
            full.release_x16_arena()
            for _ in range(2):  # the arena is sized from the previous step's need: the second step reuses the casts
                loss, g, _ = ragged_step(full, samples)
            got[keep] = (g, loss)
            if keep:  # round 4: the arena is a torch tensor of the module (sola_set_x16_arena), 1/8 above what the first step asked for
                assert full.x16_arena_bytes() > 0
                held = full.x16_arena_bytes()
                full.x16_arena_max_bytes = 1 << 20  # a cap below the need: the arena is not grown past it, the backward casts what did not fit
                full.release_x16_arena()
                for _ in range(2):
                    loss_c, g_c, _ = ragged_step(full, samples)
                assert 0 < full.x16_arena_bytes() <= (1 << 20) < held
                assert torch.equal(loss_c, loss) and all(torch.equal(g_c[k], g[k]) for k in g)
                full.x16_arena_max_bytes = None
            else:
                assert full.x16_arena_bytes() == 0  # nothing is kept, nothing is asked for
    finally:
        full.precision = "f32"
        _lib.check(_lib.lib().sola_tune(b"train_x16_keep", 1), "tune")
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 3), "tune")
    assert torch.equal(got[1][1], got[0][1])
    bad = [k for k in got[1][0] if not torch.equal(got[1][0][k], got[0][0][k])]
    assert not bad, bad


def test_bf16_statistics_pass_that_is_the_cast_gives_the_same_ragged_gradients_bit_for_bit(full):
    """Round 5 (tests/test_gpu_backward.py has the uniform-batch twin): in the bf16 step the pass over a gradient matrix that takes its
    bias sums writes its unscaled bf16 cast too (sola_tune "bwd_fused_bf16_cast" 1, the default) - against the two-pass path (max|x| +
    sums, then power-of-two-scaled casts): the same loss, every gradient bit for bit, on a ragged batch.  (With sola_tune
    "train_bf16_store" 0: round 6's storage mode takes the bias sums from the producers' rounded bf16 rows.)"""
    cfg = synth.DEFAULT_MODEL_CFG
    full.precision = "bf16"
    got = {}
    try:
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 0), "tune")
        samples = [sample_inputs(cfg, N, T, L, 450 + i) for i, (N, T, L) in enumerate(FULL_SHAPES)]
        for fused in (1, 0):
            _lib.check(_lib.lib().sola_tune(b"bwd_fused_bf16_cast", fused), "tune")
            loss, g, _ = ragged_step(full, samples)
            got[fused] = (g, loss)
    finally:
        full.precision = "f32"
        _lib.check(_lib.lib().sola_tune(b"bwd_fused_bf16_cast", 1), "tune")
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 3), "tune")
    assert torch.equal(got[1][1], got[0][1])
    bad = [k for k in got[1][0] if not torch.equal(got[1][0][k], got[0][0][k])]
    assert not bad, bad


@pytest.mark.parametrize("precision", ["f16x3", "f16", "bf16"])
@pytest.mark.parametrize("train", [True, False])
def test_attention_written_operand_casts_give_the_same_ragged_step_bit_for_bit(full, precision, train):
    """Round 4: the training forward's attention kernels write the out-projection's operand cast themselves (AttnDesc::o_cast; sola_tune
    "train_attn_cast" 1, the default) - split pairs + the plain-f16 side copy in the split-f16 step, f16 / bf16 rows in the 16-bit operand
    steps.  Against the separate cast launch (0): same loss, every gradient bit for bit, with dropout on (training mode: the simple
    kernel's training instantiation takes all three attentions of a ragged batch) and off."""
    cfg = synth.DEFAULT_MODEL_CFG
    full.precision = precision
    was_training = full.training
    full.train(train)
    got = {}
    try:
        # (bf16, train_bf16_store 3: where the attention kernel writes the bf16 rows itself it writes ONLY them, and D = dO . O sees the rounded
        # rows - a different, equally valid step; this A/B is about the cast's route, so it runs at level 2)
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 2), "tune")
        samples = [sample_inputs(cfg, N, T, L, 450 + i) for i, (N, T, L) in enumerate(FULL_SHAPES)]
        for fold in (2, 0):  # 2: the split-f16 step too (1, the default, covers the f16 / bf16 operand steps only)
            _lib.check(_lib.lib().sola_tune(b"train_attn_cast", fold), "tune")
            torch.manual_seed(1234)  # the dropout masks' seeds come from torch's CPU generator: the same two masks for both settings
            for _ in range(2):
                loss, g, _ = ragged_step(full, samples)
            got[fold] = (g, loss)
    finally:
        full.precision = "f32"
        full.train(was_training)
        _lib.check(_lib.lib().sola_tune(b"train_attn_cast", 1), "tune")
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 3), "tune")
    assert torch.equal(got[2][1], got[0][1])
    bad = [k for k in got[2][0] if not torch.equal(got[2][0][k], got[0][0][k])]
    assert not bad, bad


@pytest.mark.parametrize("precision", ["f16x3", "f16", "bf16"])
def test_row_major_weight_gradient_route_equals_the_transposed_copy_route_on_a_ragged_batch(full, precision):
    """sola_tune "train_tn_tr" 1 (default: gemm_tn_tr_kernel on row-major 16-bit operands, conv taps gathered through the ragged row
    maps in the DMA addresses, the forward's kept operand casts) against 0 (transposing casts, one per conv tap, + the NT kernel): the
    same 16-bit operand values enter both, only the f32 accumulation order differs - every gradient tensor within 2e-5 of its norm
    (a wrong tap, row map entry or zero-padding decision would show at the 1e-2 level)."""
    cfg = synth.DEFAULT_MODEL_CFG
    full.precision = precision
    got = {}
    try:
        # (round 6: the bf16 step's q / k / v storage exists on the row-major route only - there is no f32 gradient to transpose - so it is
        # switched off for this A/B of the two dW routes on the SAME operand values)
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 0), "tune")
        samples = [sample_inputs(cfg, N, T, L, 500 + i) for i, (N, T, L) in enumerate(FULL_SHAPES)]
        for route in (1, 0):
            _lib.check(_lib.lib().sola_tune(b"train_tn_tr", route), "tune")
            for _ in range(2):
                loss, g, _ = ragged_step(full, samples)
            got[route] = g
    finally:
        full.precision = "f32"
        _lib.check(_lib.lib().sola_tune(b"train_tn_tr", 1), "tune")
        _lib.check(_lib.lib().sola_tune(b"train_bf16_store", 3), "tune")
    worst = max((float((got[1][k].double() - got[0][k].double()).norm()) / (float(got[0][k].double().norm()) + 1e-30), k) for k in got[0])
    print("row-major vs transposed-copy dW route, worst tensor:", worst)
    assert worst[0] < 2e-5, worst


def test_full_gradient_norms_of_single_golden_samples(full_golden, full):
    """A ragged batch of ONE golden sample must reproduce the reference's per-parameter gradient norms (the ragged kernels
    alone, no summation over samples)."""
    cfg = synth.DEFAULT_MODEL_CFG
    for ci in range(len(full_golden["cases"])):
        B, N, T, L = [int(v) for v in full_golden["cases"][ci]]
        if B != 1 or N * T > 4096:
            continue
        g = case_dict(full_golden, ci)
        s = sample_inputs(cfg, N, T, L, 200 + ci)
        loss, grads, _ = ragged_step(full, [s])
        np.testing.assert_allclose(loss[0].cpu().numpy().astype(np.float64), g["loss"], rtol=3e-4, atol=3e-4)
        total = float(dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norm_vals"]))["total_grad_norm"])
        bad = {}
        for k, gr in grads.items():
            ref = float(g["gradnorm." + k])
            got = float(gr.double().norm())
            if abs(got - ref) > 2e-3 * ref + 1e-5 * total:
                bad[k] = (got, ref)
        assert not bad, f"case {ci}: {bad}"


def test_equal_shape_ragged_batch_reproduces_the_uniform_batch_under_dropout(small):
    """Training mode: the dropout masks are a function of (seed, element index); an equal-shape ragged batch has the uniform
    batch's element indices, so the same seed must give the same outputs and gradients through the table-driven kernels."""
    cfg = synth.SMALL_MODEL_CFG
    B, N, T, L = 3, 6, 40, 7
    inp = synth.make_inputs(cfg, B, N, T, L, 7)
    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    small.train()
    try:
        torch.manual_seed(123)
        small.zero_grad(set_to_none=True)
        sm, st = small(c["object_tokens"], c["lang_tokens"])
        seed_u = small._last_dropout_seed
        loss3 = track_selection_losses(sm, st, c["labels"], c["pos_tokens"], small.negative_token.weight, POS_W, TEMP, ALIGN_W)
        loss3[0].backward()
        ref = {k: p.grad.detach().clone() for k, p in small.named_parameters()}
        ref_sm = sm.detach().clone()
        torch.manual_seed(123)
        small.zero_grad(set_to_none=True)
        small.forward_ragged(list(c["object_tokens"]), list(c["lang_tokens"]))
        assert small._last_dropout_seed == seed_u and seed_u != 0
        flat, tok, offs, counts = small.last_ragged
        # the uniform loss is ONE mean over all B*N tracks = the mean of the per-sample means here (equal track counts)
        loss = track_selection_losses_ragged(flat, tok, c["labels"].reshape(-1), c["pos_tokens"][:, 0], small.negative_token.weight,
                                             offs, counts, POS_W, TEMP, ALIGN_W)
        loss[:, 0].mean().backward()
        got = {k: p.grad.detach().clone() for k, p in small.named_parameters()}
        # eval-mode outputs differ: the masks really are on
        small.eval()
        with torch.no_grad():
            sm_eval, _ = small(c["object_tokens"], c["lang_tokens"])
        assert float((sm_eval - ref_sm).abs().max()) > 1e-3
    finally:
        small.eval()
    torch.testing.assert_close(flat.detach().reshape(B, N), ref_sm, rtol=0, atol=2e-4)
    assert_grads_close(got, ref, 2e-3, "dropout, ragged vs uniform")


def test_call_sequence_errors(small):
    cfg = synth.SMALL_MODEL_CFG
    s = sample_inputs(cfg, 4, 16, 5, 1)
    lib = _lib.lib()
    # a ragged forward, then the uniform backward entry point (and the other way round) is a state error, not a crash
    small.zero_grad(set_to_none=True)
    small.forward_ragged([s["obj"]], [s["lang"]], differentiable=True)
    dsm = torch.zeros(4, device="cuda")
    dst = torch.zeros(4, cfg["lang_token_dim"], device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    st = lib.sola_backward(small._ctx, _lib.ptr(dsm), _lib.ptr(dst), _lib.ptr(small._train_ws), _lib.ptr(ws), ws.numel(), None)
    assert st == -5, st
    # a different workspace than the forward's
    other = torch.empty_like(small._train_ws)
    st = lib.sola_backward_ragged(small._ctx, _lib.ptr(dsm), _lib.ptr(dst), _lib.ptr(other), _lib.ptr(ws), ws.numel(), None)
    assert st == -5, st
    # shared videos are an inference feature; in training every sample is its own video (the binding repeats the tokens)
    sms, _ = small.forward_ragged([s["obj"]], [s["lang"], s["lang"]], [0, 0], differentiable=True)
    assert len(sms) == 2 and sms[0].requires_grad
    torch.testing.assert_close(sms[0], sms[1], rtol=0, atol=0)
    with pytest.raises(SolaError):
        small.forward_ragged([s["obj"]], [s["lang"]], [1])


def test_dropout_gradients_of_a_mixed_shape_batch_match_finite_differences(small):
    """Training mode on a MIXED-shape batch: the dropout masks are functions of (seed, element index / unit index), regenerated by
    the backward kernels from the unit tables.  With the seed pinned the loss is a smooth function of the parameters, so central
    differences of the loss must agree with the HIP gradients - for parameters upstream of every dropout site (encoder norms'
    dropout, the three attentions' dropout on the probabilities)."""
    cfg = synth.SMALL_MODEL_CFG
    shapes = [(6, 40, 7), (3, 9, 2), (17, 130, 5), (9, 24, 11)]
    samples = [sample_inputs(cfg, N, T, L, 900 + i) for i, (N, T, L) in enumerate(shapes)]
    labels = torch.cat([s["labels"] for s in samples])
    pos = torch.stack([s["pos"] for s in samples])
    small.train()
    try:
        def loss_value(backward):
            torch.manual_seed(4321)  # the step's dropout seed is drawn from torch's generator
            small.zero_grad(set_to_none=True)
            small.forward_ragged([s["obj"] for s in samples], [s["lang"] for s in samples])
            assert small._last_dropout_seed != 0
            flat, tok, offs, counts = small.last_ragged
            loss = track_selection_losses_ragged(flat, tok, labels, pos, small.negative_token.weight, offs, counts, POS_W, TEMP, ALIGN_W)[:, 0].sum()
            if backward:
                loss.backward()
            return float(loss.detach().double())

        base = loss_value(True)
        named = dict(small.named_parameters())
        grads = {k: p.grad.detach().clone() for k, p in named.items()}
        assert loss_value(False) == base  # same seed -> same masks -> bit-identical loss
        probes = [("short_motion_encoder.0.bias", (3,)), ("short_motion_encoder.5.weight", (10,)), ("short_motion_encoder.12.bias", (7,)),
                  ("object_lang_align_layers.0.obj_attn.v_proj.bias", (11,)), ("object_lang_align_layers.0.motion_attn.out_proj.bias", (2,)),
                  ("object_lang_align_layers.1.object2lang_attn.q_proj.bias", (5,)), ("object_lang_align_layers.1.norm.1.weight", (9,)),
                  ("negative_token.weight", (1, 4))]
        bad = {}
        for key, idx in probes:
            p = named[key]
            eps = 2e-2
            with torch.no_grad():
                old = float(p[idx])
                p[idx] = old + eps
            small.weights_changed()
            up = loss_value(False)
            with torch.no_grad():
                p[idx] = old - eps
            small.weights_changed()
            dn = loss_value(False)
            with torch.no_grad():
                p[idx] = old
            small.weights_changed()
            fd = (up - dn) / (2 * eps)
            g = float(grads[key][idx])
            if abs(fd - g) > 3e-2 * max(abs(g), abs(fd)) + 2e-3:
                bad[key] = (fd, g)
        assert not bad, f"finite differences vs HIP gradients under dropout: {bad}"
    finally:
        small.eval()


@pytest.mark.parametrize("shapes", [[(1, 1, 1)], [(1, 9, 1), (130, 24, 3), (5, 300, 7)], [(97, 8, 2), (2, 2, 2), (40, 257, 33)]])
def test_edge_shapes_equal_one_sample_steps(small, shapes):
    """One sample of one track, one frame, one text token; more than 96 / 128 tracks (the block-shared attention backward on unit
    tables instead of the per-wave ragged form; inter-object units past two 64-row blocks); more than 32 encoded steps."""
    cfg = synth.SMALL_MODEL_CFG
    samples = [sample_inputs(cfg, N, T, L, 700 + i) for i, (N, T, L) in enumerate(shapes)]
    ref_sum, ref_loss = None, []
    for s in samples:
        l3, g, _ = one_sample_step(small, s)
        ref_loss.append(l3)
        ref_sum = g if ref_sum is None else {k: ref_sum[k] + g[k] for k in g}
    loss, got, _ = ragged_step(small, samples)
    torch.testing.assert_close(loss, torch.stack(ref_loss), rtol=2e-4, atol=2e-4)
    assert_grads_close(got, ref_sum, 2e-3, f"edge shapes {shapes}")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_training_forward_of_the_benched_ragged_batch_every_logit_vs_oracle(full, precision):
    """the 64-sample ragged TRAINING batch of bench.py (synth.make_ragged_samples seed 2024) through
    sola_forward_train_ragged (differentiable call, eval mode: dropout off) - EVERY logit against the fp32 oracle's per-sample forward,
    in exact f32 and with the split-f16 GEMMs: the north star's 1e-3, selections equal."""
    from oracle import sola_oracle  # checker only

    cfg = synth.DEFAULT_MODEL_CFG
    smp = synth.make_ragged_samples(cfg, 64, 2024)
    tsd = sola_oracle.to_torch_state(synth.make_state_dict(cfg, 42))
    torch.set_num_threads(32)
    ref = np.concatenate([np.asarray(sola_oracle.forward(tsd, cfg, x["obj"].numpy()[None], x["lang"].numpy()[None])[0])[0] for x in smp])
    full.eval()
    full.precision = precision
    try:
        full.forward_ragged([x["obj"].cuda() for x in smp], [x["lang"].cuda() for x in smp], differentiable=True)
        flat, _tok, _offs, counts = full.last_ragged
        assert flat.requires_grad
        got = flat.detach().cpu().numpy()
    finally:
        full.precision = "f32"
    e = np.abs(got - ref)
    starts = np.cumsum([0] + list(counts[:-1]))
    per = np.array([e[o:o + c].max() for o, c in zip(starts, counts)])
    print(f"training forward {precision}: worst logit error {e.max():.3e}, mean per-sample worst {per.mean():.3e}, samples > 5e-4: {(per > 5e-4).sum()} of 64")
    np.testing.assert_array_equal(got > 0, ref > 0)
    assert e.max() <= 1e-3
    # round 6: against the REFERENCE's own logits for this batch (tests/golden/gen_golden.py bench)
    gold = _load("bench_golden.npz")
    assert list(gold["rag_train.2024.counts"]) == list(counts)
    er = np.abs(got - gold["rag_train.2024.score_map"])
    print(f"training forward {precision} vs REFERENCE: worst logit error {er.max():.3e}, mean {er.mean():.3e}")
    np.testing.assert_array_equal((torch.sigmoid(torch.from_numpy(got)) > 0.5).numpy(), gold["rag_train.2024.selected"])
    assert er.max() <= 1e-3
