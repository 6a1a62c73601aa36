"""Split-f16 precision mode (module.precision = "f16x3"): the convs and projections run as three f16 MFMAs per product
on (f16 hi, f16 lo) operand pairs with f32 accumulation.  It must satisfy the SAME parity bar as the exact-f32 mode:
logits/tokens within the north-star 1e-3 of the reference's golden vectors, bit-exact selections / arg-max / hardest
negatives, plus kernel-level checks of the split representation (22-bit products) against float64."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import case_dict  # noqa: E402
from oracle import sola_oracle  # noqa: E402
from sola_amd import ops, synth  # noqa: E402
from sola_amd.loss import track_selection_losses  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

POS_W, TEMP, ALIGN_W = 1.5, 0.07, 0.3


def cuda(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def test_split_representation_has_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 256)) * np.exp(rng.uniform(-6, 3, size=(64, 256)))).astype(np.float32)
    for scale in (1.0, 64.0):
        dec = ops.decode_sp16(ops.cast_sp16(cuda(x), scale)).cpu().numpy().astype(np.float64)
        v = x.astype(np.float64) * scale
        err = np.abs(dec - v)
        # hi carries 11 bits, lo the next 11; once lo drops below the f16 normal range (|v| < ~0.1) its absolute
        # resolution is the subnormal spacing 2^-24, i.e. an error floor of 3e-8 regardless of |v|
        assert np.all(err <= 2.0 ** -21 * np.abs(v) + 3.1e-8), float((err - 2.0 ** -21 * np.abs(v)).max())
        assert np.abs(dec[np.abs(v) > 0.25] / v[np.abs(v) > 0.25] - 1).max() < 2.0 ** -21


@pytest.mark.parametrize("M,N,K", [(256, 1024, 1024), (100, 72, 96), (4096, 512, 768), (16384, 1024, 1024), (48, 2048, 1024)])
def test_split_gemm_vs_float64(M, N, K):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.uniform(-1, 1, size=(N, K)) / 32).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    r = rng.standard_normal((M, N)).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T + b + r
    got = ops.gemm_nt_split(ops.cast_sp16(cuda(a)), ops.cast_sp16(cuda(w), 64.0), cuda(b), cuda(r), False, 1.0 / 64.0).cpu().numpy()
    f32 = ops.gemm_nt(cuda(a), cuda(w), cuda(b), cuda(r)).cpu().numpy()
    err_split = np.abs(got - ref).max()
    err_f32 = np.abs(f32 - ref).max()
    # same error class as exact-f32 accumulation (f32 accumulate dominates; products carry 22 bits)
    assert err_split <= max(4 * err_f32, 2e-6 * np.abs(ref).max()), (err_split, err_f32)
    # residual given in the split format
    got2 = ops.gemm_nt_split(ops.cast_sp16(cuda(a)), ops.cast_sp16(cuda(w), 64.0), cuda(b), ops.cast_sp16(cuda(r[:, : (N // 8) * 8]))
                             if N % 8 == 0 else cuda(r), N % 8 == 0, 1.0 / 64.0).cpu().numpy()
    assert np.abs(got2 - ref).max() <= max(4 * err_f32, 3e-6 * np.abs(ref).max())


def build(cfg, precision):
    m = LanguageAlignedTrackSelectionModule(cfg)
    sd = synth.make_state_dict(cfg, 42)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    return m, sd


def run(m, cfg, B, N, T, L, seed):
    inp = synth.make_inputs(cfg, B, N, T, L, seed)
    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    with torch.no_grad():
        sm, st = m(c["object_tokens"], c["lang_tokens"])
        loss3, argmax = track_selection_losses(sm, st, c["labels"], c["pos_tokens"], m.negative_token.weight, POS_W, TEMP, ALIGN_W,
                                               return_argmax=True)
    return inp, sm, st, loss3, argmax


@pytest.fixture(scope="module")
def full_fast():
    return build(synth.DEFAULT_MODEL_CFG, "f16x3")


@pytest.fixture(scope="module")
def full_f32():
    return build(synth.DEFAULT_MODEL_CFG, "f32")


@pytest.mark.parametrize("ci", range(5))
def test_full_cases_vs_golden_in_split_mode(full_golden, full_fast, ci):
    m, _ = full_fast
    cfg = synth.DEFAULT_MODEL_CFG
    B, N, T, L = [int(v) for v in full_golden["cases"][ci]]
    g = case_dict(full_golden, ci)
    _, sm, st, loss3, argmax = run(m, cfg, B, N, T, L, 200 + ci)
    assert np.abs(sm.cpu().numpy() - g["score_map"]).max() <= 1e-3
    assert np.abs(st.cpu().numpy() - g["score_tokens"]).max() <= 1e-3
    np.testing.assert_allclose(loss3.cpu().numpy().astype(np.float64), g["loss"], rtol=2e-4, atol=2e-4)
    np.testing.assert_array_equal((torch.sigmoid(sm) > 0.5).float().cpu().numpy(), g["selected"])
    np.testing.assert_array_equal(sm.argmax(dim=1).cpu().numpy(), g["argmax_track"])
    np.testing.assert_array_equal(argmax.cpu().numpy(), g["neg_argmax"])


def test_split_mode_error_is_in_the_f32_class(full_fast, full_f32):
    """Against a float64 evaluation of the oracle at the north-star shape: the split mode's error is no worse than
    twice the exact-f32 mode's (both are dominated by f32 accumulation / GroupNorm rounding)."""
    cfg = synth.DEFAULT_MODEL_CFG
    mf, sd = full_fast
    m32, _ = full_f32
    inp, sm_f, st_f, _, _ = run(mf, cfg, 2, 64, 32, 16, 4321)
    _, sm_3, st_3, _, _ = run(m32, cfg, 2, 64, 32, 16, 4321)
    rsm, rst = sola_oracle.forward(sd, cfg, inp["object_tokens"], inp["lang_tokens"], dtype=torch.float64)
    e_fast = max(np.abs(sm_f.cpu().numpy() - rsm.numpy()).max(), np.abs(st_f.cpu().numpy() - rst.numpy()).max())
    e_f32 = max(np.abs(sm_3.cpu().numpy() - rsm.numpy()).max(), np.abs(st_3.cpu().numpy() - rst.numpy()).max())
    print(f"max error vs float64: split-f16 {e_fast:.3e}, exact f32 {e_f32:.3e}")
    assert e_fast <= 5e-4 and e_fast <= 2.0 * e_f32 + 5e-5


@pytest.mark.parametrize("ci", range(6))
def test_small_cases_in_split_mode(small_golden, ci):
    cfg = synth.SMALL_MODEL_CFG
    m, _ = build(cfg, "f16x3")
    B, N, T, L = [int(v) for v in small_golden["cases"][ci]]
    g = case_dict(small_golden, ci)
    _, sm, st, loss3, argmax = run(m, cfg, B, N, T, L, 100 + ci)
    assert np.abs(sm.cpu().numpy() - g["score_map"]).max() <= 1e-3
    assert np.abs(st.cpu().numpy() - g["score_tokens"]).max() <= 1e-3
    np.testing.assert_array_equal((torch.sigmoid(sm) > 0.5).float().cpu().numpy(), g["selected"])
    np.testing.assert_array_equal(argmax.cpu().numpy(), g["neg_argmax"])
    # intermediate in the split format decodes to the reference activations
    ref = g["tap.l1_motion"]
    got = ops.decode_sp16(m.workspace_tap("l1_motion")).cpu().numpy().reshape(ref.shape)
    assert np.abs(got - ref).max() <= 3e-4 * max(1.0, np.abs(ref).max())
