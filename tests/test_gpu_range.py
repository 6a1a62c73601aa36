"""Range safety of the split-f16 mode (module.precision = "f16x3"), against outputs of the REAL reference
(tests/golden/range_golden.npz, made by tests/golden/gen_golden.py from module/module.py:130-162):

* object / text tokens scaled by 1e-5 ... 1e3: the data-dependent power-of-two scales keep the split path itself inside
  the parity bar - no fallback is allowed to hide a failure here;
* projection weights scaled, shrunk or carrying outliers, conv weights x100: per-matrix device-side scales, no fallback;
* GroupNorm weights that put activations outside what the fixed activation scale covers, or a value beyond the f16 range:
  the guard trips and the call is repeated on the exact-f32 kernels (bit-identical to the f32 mode);
* a 256-sample batch at the headline shape, sampled rows against the oracle.

Tolerance: the north-star 1e-3 on logits of the usual magnitude (|score| <= 16); where the scaled inputs / weights make the
logits themselves large the same RELATIVE bar is used (1e-3 / 16 of the largest reference logit).  Two variants (weights
x8, GroupNorm x300 / x3000) saturate the softmaxes into arg-maxes: there the reference's own fp32 result sits up to 0.1
from a float64 evaluation (stored as "cond" next to each golden output) and the bar is 3x that distance - those cases
check that nothing overflows or falls apart, not the fourth digit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import _load  # noqa: E402
from oracle import sola_oracle  # noqa: E402
from sola_amd import synth  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

CFG = synth.DEFAULT_MODEL_CFG


@pytest.fixture(scope="module")
def gold():
    return _load("range_golden.npz")


def build(variant, precision):
    m = LanguageAlignedTrackSelectionModule(CFG)
    sd = synth.make_state_dict_variant(CFG, 42, variant)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    return m, sd


@pytest.fixture(scope="module")
def base_models():
    return {p: build("base", p)[0] for p in ("f16x3", "f32")}


def forward(m, gold, so=1.0, sl=1.0):
    B, N, T, L = [int(v) for v in gold["shape"]]
    inp = synth.make_inputs(CFG, B, N, T, L, seed=300)
    obj = torch.from_numpy(inp["object_tokens"] * np.float32(so)).cuda()
    lang = torch.from_numpy(inp["lang_tokens"] * np.float32(sl)).cuda()
    with torch.no_grad():
        sm, st = m(obj, lang)
    return sm.cpu().numpy(), st.cpu().numpy()


def check(sm, st, g_sm, g_st, what, cond=(0.0, 0.0)):
    tol_sm = max(1e-3 * max(1.0, float(np.abs(g_sm).max()) / 16.0), 3.0 * float(cond[0]))
    tol_st = max(1e-3 * max(1.0, float(np.abs(g_st).max()) / 16.0), 3.0 * float(cond[1]))
    e_sm = float(np.abs(sm - g_sm).max())
    e_st = float(np.abs(st[:, :8] - g_st).max())
    print(f"{what}: |score_map err| {e_sm:.3e} (tol {tol_sm:.1e}), |score_tokens err| {e_st:.3e} (tol {tol_st:.1e})")
    assert e_sm <= tol_sm and e_st <= tol_st, (what, e_sm, tol_sm, e_st, tol_st)
    # decisions: identical wherever the reference logit is not within the tolerance of the threshold
    clear = np.abs(g_sm) > 2 * tol_sm
    np.testing.assert_array_equal((sm > 0)[clear], (g_sm > 0)[clear])


@pytest.mark.parametrize("i", range(len(synth.RANGE_INPUT_SCALES)))
def test_input_scales_split_path_meets_the_bar_without_fallback(gold, base_models, i):
    so, sl = synth.RANGE_INPUT_SCALES[i]
    assert tuple(gold["input_scales"][i]) == (so, sl)
    m = base_models["f16x3"]
    n0, _ = m.split_fallbacks()
    sm, st = forward(m, gold, so, sl)
    n1, bits = m.split_fallbacks()
    assert n1 == n0 and bits == 0, f"the split path itself must hold at token scales {(so, sl)} (guard bits {bits})"
    cond = gold[f"in{i}.cond"]
    assert cond.max() < 1e-2  # the input-scale cases are well-conditioned: the plain bar applies (3 x cond matters only at (1, 16))
    check(sm, st, gold[f"in{i}.score_map"], gold[f"in{i}.score_tokens"], f"f16x3 token scales {(so, sl)}", cond)
    sm32, st32 = forward(base_models["f32"], gold, so, sl)
    check(sm32, st32, gold[f"in{i}.score_map"], gold[f"in{i}.score_tokens"], f"f32   token scales {(so, sl)}", cond)


@pytest.mark.parametrize("variant", [v for v in synth.WEIGHT_VARIANTS])
def test_weight_variants(gold, variant):
    m, _ = build(variant, "f16x3")
    sm, st = forward(m, gold)
    n, bits = m.split_fallbacks()
    check(sm, st, gold[f"w.{variant}.score_map"], gold[f"w.{variant}.score_tokens"], f"f16x3 weights {variant}", gold[f"w.{variant}.cond"])
    if variant in ("gamma_div256", "gamma_x3000"):
        # GroupNorm outputs of rms ~0.004 / ~3000: outside what the fixed activation scale covers -> exact-f32 kernels
        assert n == 1 and (bits & 2), (variant, n, bits)
        m32, _ = build(variant, "f32")
        sm32, st32 = forward(m32, gold)
        np.testing.assert_array_equal(sm, sm32)
        np.testing.assert_array_equal(st, st32)
    elif variant != "gamma_x300":
        assert n == 0 and bits == 0, f"{variant}: the per-matrix scales must keep the split path in range (guard bits {bits})"


def test_value_beyond_f16_range_trips_the_guard_and_the_call_is_repeated_in_f32(gold):
    """One conv5 bias channel at 1e5: conv5's output is written as split-f16 pairs, whose hi half would be inf."""
    outs = {}
    for prec in ("f16x3", "f32"):
        m, _ = build("base", prec)
        with torch.no_grad():
            m.short_motion_encoder[20].bias[3] = 1.0e5
        outs[prec] = forward(m, gold)
        if prec == "f16x3":
            n, bits = m.split_fallbacks()
            assert n == 1 and (bits & 1), (n, bits)
            m.split_guard = False  # fully asynchronous call: the result is the split path's, the guard is not consulted
            sm_u, _ = forward(m, gold)
            assert m.split_fallbacks()[0] == 1
            assert not np.isfinite(sm_u).all() or np.abs(sm_u - outs["f16x3"][0]).max() > 0  # it really was out of range
    assert np.isfinite(outs["f32"][0]).all()
    np.testing.assert_array_equal(outs["f16x3"][0], outs["f32"][0])
    np.testing.assert_array_equal(outs["f16x3"][1], outs["f32"][1])


def test_weights_changed_rebuilds_scales_and_rechecks(gold):
    """In-place weight updates (optimizer step, load_state_dict) must refresh the per-matrix scales and the norm check."""
    m, sd = build("base", "f16x3")
    forward(m, gold)
    assert m.split_fallbacks() == (0, 0)
    big = synth.make_state_dict_variant(CFG, 42, "lin_x8")
    with torch.no_grad():
        for k, p in m.state_dict(keep_vars=True).items():
            p.copy_(torch.from_numpy(big[k]))
    sm, st = forward(m, gold)
    check(sm, st, gold["w.lin_x8.score_map"], gold["w.lin_x8.score_tokens"], "f16x3 after in-place x8", gold["w.lin_x8.cond"])
    assert m.split_fallbacks() == (0, 0)
    small = synth.make_state_dict_variant(CFG, 42, "gamma_div256")
    with torch.no_grad():
        for k, p in m.state_dict(keep_vars=True).items():
            p.copy_(torch.from_numpy(small[k]))
    sm, st = forward(m, gold)
    check(sm, st, gold["w.gamma_div256.score_map"], gold["w.gamma_div256.score_tokens"], "f16x3 after in-place gamma/256")
    n, bits = m.split_fallbacks()
    assert n == 1 and (bits & 2)


_ORACLE_CACHE = {}


@pytest.fixture(scope="module")
def bench_gold():
    """tests/golden/bench_golden.npz: the REFERENCE's own logits (module/module.py:130-162 called one sample per forward, as
    inference.py:58 does) for every batch bench.py times - tests/golden/gen_golden.py bench."""
    return _load("bench_golden.npz")


def _vs_reference(sm, ref_flat, sel_flat, what):
    """every row of a batch against the reference's own logits: worst / mean row error, the thresholded selection (inference.py:59-60)"""
    ref = ref_flat.reshape(sm.shape)
    e_rows = np.abs(sm - ref).max(axis=1)
    print(f"{what} vs REFERENCE: logits worst {e_rows.max():.3e} mean {e_rows.mean():.3e} rows>5e-4 {(e_rows > 5e-4).sum()}")
    np.testing.assert_array_equal((torch.sigmoid(torch.from_numpy(sm)) > 0.5).numpy(), sel_flat.reshape(sm.shape))
    return e_rows


def _oracle_rows(seed, B, N, T, L):
    """fp32 oracle logits / tokens of EVERY row of a batch (computed once per seed; ~8 s of host time at B = 256)."""
    key = (seed, B, N, T, L)
    if key not in _ORACLE_CACHE:
        inp = synth.make_inputs(CFG, B, N, T, L, seed=seed)
        tsd = sola_oracle.to_torch_state(synth.make_state_dict(CFG, 42))
        sms, sts = [], []
        for b in range(0, B, 16):
            rsm, rst = sola_oracle.forward(tsd, CFG, inp["object_tokens"][b:b + 16], inp["lang_tokens"][b:b + 16])
            sms.append(rsm.numpy()); sts.append(rst.numpy())
        _ORACLE_CACHE[key] = (inp, np.concatenate(sms), np.concatenate(sts))
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("seed", [1000, 1001, 1002])
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_batch_256_every_row_vs_oracle(base_models, bench_gold, precision, seed):
    """The benched batch (seed 1000 = bench.py's rank-0 batch) and two more: 256 samples at (T=32, N=64, L=16) in one call, EVERY
    row against the per-sample fp32 oracle (six sampled rows of one seed left no margin).  The bar: north star 1e-3 on
    every logit and token, thresholded selections equal, and - the margin - a mean row error <= 2.5e-4 with at most 3 % of the rows
    above 5e-4.  A worst row below 5e-4 is not attainable against an fp32 checker: measured against a float64 evaluation
    (tools/batch_error.py, profiles/r03_batch_errors.jsonl) the fp32 oracle - the reference's own arithmetic - has a worst row of
    5.5e-4 on this batch (mean 1.5e-4); the exact-f32 kernels sit at 1.4e-4 mean / 9.1e-4 worst, the split-f16 mode at 1.9e-4 /
    7.3e-4: the tail is a few ill-conditioned rows (saturated first inter-object softmax), where two correct fp32 evaluation
    orders differ by the sum of their own errors."""
    B, N, T, L = 256, 64, 32, 16
    inp, rsm, rst = _oracle_rows(seed, B, N, T, L)
    m = base_models[precision]
    with torch.no_grad():
        sm, st = m(torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda())
    sm, st = sm.cpu().numpy(), st.cpu().numpy()
    assert m.split_fallbacks()[1] == 0
    e_rows = np.abs(sm - rsm).max(axis=1)
    e_tok = float(np.abs(st - rst).max())
    print(f"{precision} seed {seed}: logits worst {e_rows.max():.3e} mean {e_rows.mean():.3e} rows>5e-4 {(e_rows > 5e-4).sum()}; tokens worst {e_tok:.3e}")
    np.testing.assert_array_equal(sm > 0, rsm > 0)
    # the north star's 1e-3, on the logits AND on the 1024-dim score tokens (entries up to ~4; 16.8 M values per batch, measured worst
    # 9.5e-4 - round 4, ADVICE r3: held to the same hard bound instead of 1.5e-3)
    assert e_rows.max() <= 1e-3 and e_tok <= 1e-3
    assert e_rows.mean() <= 2.5e-4 and (e_rows > 5e-4).sum() <= 8
    # round 6: the same rows against the REFERENCE's own fp32 logits - the north-star sentence itself.  Two correct
    # fp32 evaluations of this network differ by the sum of their own distances from exact arithmetic (reference 4.9e-4 worst, 1.4e-4 mean
    # on seed 1000 against float64, bench_golden "u256.1000.oracle_f64"), so the margin here is smaller than against the oracle.
    r_rows = _vs_reference(sm, bench_gold[f"u256.{seed}.score_map"], bench_gold[f"u256.{seed}.selected"], f"{precision} seed {seed}")
    assert r_rows.max() <= 1e-3 and r_rows.mean() <= 2.5e-4
    if seed == 1000:  # and against exact arithmetic: no further from it than twice the reference itself
        x_rows = np.abs(sm - bench_gold["u256.1000.oracle_f64.score_map"].reshape(sm.shape)).max(axis=1)
        ref_x = np.abs(bench_gold["u256.1000.score_map"].astype(np.float64) - bench_gold["u256.1000.oracle_f64.score_map"]).reshape(sm.shape).max(axis=1)
        print(f"{precision} vs float64: worst {x_rows.max():.3e} mean {x_rows.mean():.3e}; the reference itself: worst {ref_x.max():.3e} mean {ref_x.mean():.3e}")
        assert x_rows.max() <= 1e-3 and x_rows.mean() <= 2 * ref_x.mean()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_batch_256_on_unsaturated_weights_vs_reference(bench_gold, precision):
    """VERDICT r5 item 2 / weak point 1: at random-init weights the first inter-object softmax is saturated (scores of rms ~100) and sets the
    error tail.  The benched batch (seed 1000) on weights whose projections are scaled by 1/64 (synth variant "lin_div64": a nearly uniform
    softmax, the regime of a trained network whose attention is not an arg-max) against the reference's own logits: every row, both modes."""
    B, N, T, L = 256, 64, 32, 16
    m, _sd = build("lin_div64", precision)
    inp = synth.make_inputs(CFG, B, N, T, L, seed=1000)
    with torch.no_grad():
        sm, _st = m(torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda())
    assert m.split_fallbacks()[1] == 0
    r_rows = _vs_reference(sm.cpu().numpy(), bench_gold["u256.1000.lin_div64.score_map"], bench_gold["u256.1000.lin_div64.selected"], f"{precision} lin_div64")
    assert r_rows.max() <= 1e-3 and r_rows.mean() <= 2.5e-4


def test_16_bit_storage_mode_vs_reference(bench_gold):
    """VERDICT r5 row g: the 16-bit STORAGE mode of inference (module.precision = "f16": every activation between two kernels a plain
    _Float16) against the REFERENCE's own logits for the benched batch.  It is a reduced-precision mode with a stated tolerance, not a
    parity mode: at random-init weights the first inter-object softmax is saturated (scores of rms ~100) and 2^-11 on q / k flips near-ties
    (measured: worst 0.52, rms 0.032 on logits of magnitude ~10, 0.15 % of them beyond 0.25); on weights whose softmax is not saturated
    ("lin_div64") what is left is the format's own rounding through 18 GEMM layers (measured: worst 0.034, rms 8.1e-3 = 8e-4 of the logits'
    magnitude).  Bounds: 1.0 / 0.04 and 0.06 / 0.012; every selection whose logit is further than twice the worst error from 0 equal."""
    B, N, T, L = 256, 64, 32, 16
    inp = synth.make_inputs(CFG, B, N, T, L, seed=1000)
    obj, lang = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
    for variant, key, worst_tol, rms_tol in (("base", "u256.1000", 1.0, 0.04), ("lin_div64", "u256.1000.lin_div64", 0.06, 0.012)):
        m, _sd = build(variant, "f16")
        with torch.no_grad():
            sm, _st = m(obj, lang)
        sm = sm.cpu().numpy()
        ref = bench_gold[key + ".score_map"].reshape(sm.shape)
        err = np.abs(sm - ref)
        rms = float(np.sqrt((err.astype(np.float64) ** 2).mean()))
        print(f"f16 storage, {variant} weights vs REFERENCE: worst {err.max():.3e} rms {rms:.3e} logits beyond 0.25: {(err > 0.25).mean():.4%}")
        assert err.max() <= worst_tol and rms <= rms_tol
        far = np.abs(ref) > 2 * max(err.max(), 1e-3)
        np.testing.assert_array_equal((sm > 0)[far], (ref > 0)[far])
        del m


def test_stress_batch_every_row_vs_oracle(base_models, bench_gold):
    """BASELINE config C4 (T=128, N=128), 32 samples in one call, every row, both modes."""
    B, N, T, L = 32, 128, 128, 16
    inp = synth.make_inputs(CFG, B, N, T, L, seed=2000)
    tsd = sola_oracle.to_torch_state(synth.make_state_dict(CFG, 42))
    ref = [sola_oracle.forward(tsd, CFG, inp["object_tokens"][b:b + 2], inp["lang_tokens"][b:b + 2]) for b in range(0, B, 2)]
    rsm = np.concatenate([r[0].numpy() for r in ref]); rst = np.concatenate([r[1].numpy() for r in ref])
    for precision in ("f16x3", "f32"):
        m = base_models[precision]
        with torch.no_grad():
            sm, st = m(torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda())
        sm, st = sm.cpu().numpy(), st.cpu().numpy()
        e_rows = np.abs(sm - rsm).max(axis=1)
        print(f"C4 {precision}: logits worst {e_rows.max():.3e} mean {e_rows.mean():.3e} tokens worst {float(np.abs(st - rst).max()):.3e}")
        np.testing.assert_array_equal(sm > 0, rsm > 0)
        assert e_rows.max() <= 1e-3 and float(np.abs(st - rst).max()) <= 1e-3 and e_rows.mean() <= 4e-4
        r_rows = _vs_reference(sm, bench_gold["c4.2000.score_map"], bench_gold["c4.2000.selected"], f"C4 {precision}")
        assert r_rows.max() <= 1e-3 and r_rows.mean() <= 4e-4
