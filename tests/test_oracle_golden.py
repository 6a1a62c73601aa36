"""Pin the CPU oracle to the golden vectors produced by the real reference (tests/golden/gen_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import case_dict
from oracle import iou_oracle, sola_oracle
from sola_amd import synth

POS_W, TEMP, ALIGN_W = 1.5, 0.07, 0.3


def _run(cfg, sd, B, N, T, L, seed, dtype=torch.float32, taps=None):
    inp = synth.make_inputs(cfg, B, N, T, L, seed)
    sm, st = sola_oracle.forward(sd, cfg, inp["object_tokens"], inp["lang_tokens"], dtype=dtype, taps=taps)
    neg = np.broadcast_to(sd["negative_token.weight"][None], (B,) + sd["negative_token.weight"].shape)
    ls = sola_oracle.losses(sm, st, inp["labels"], inp["pos_tokens"], neg, POS_W, TEMP, ALIGN_W, dtype=dtype)
    return sm, st, ls


@pytest.fixture(scope="module")
def small_sd():
    return synth.make_state_dict(synth.SMALL_MODEL_CFG, 42)


@pytest.fixture(scope="module")
def full_sd():
    return synth.make_state_dict(synth.DEFAULT_MODEL_CFG, 42)


@pytest.mark.parametrize("ci", range(6))
def test_small_forward_taps_losses(small_golden, small_sd, ci):
    B, N, T, L = [int(v) for v in small_golden["cases"][ci]]
    g = case_dict(small_golden, ci)
    taps = {}
    sm, st, ls = _run(synth.SMALL_MODEL_CFG, small_sd, B, N, T, L, 100 + ci, taps=taps)
    assert sum(("tap." + k) in g for k in taps) >= 13
    for k, v in taps.items():
        if "tap." + k not in g:  # "encoder" duplicates conv5
            continue
        ref = g["tap." + k]
        np.testing.assert_allclose(v.numpy(), ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)
    np.testing.assert_allclose(sm.numpy(), g["score_map"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(st.numpy(), g["score_tokens"], rtol=0, atol=1e-4)
    np.testing.assert_allclose([float(ls["total"]), float(ls["bce"]), float(ls["align"])], g["loss"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(ls["neg_argmax"].numpy(), g["neg_argmax"])
    np.testing.assert_array_equal(sola_oracle.select(sm).numpy(), g["selected"])
    np.testing.assert_array_equal(sm.argmax(dim=1).numpy(), g["argmax_track"])


@pytest.mark.parametrize("ci", range(5))
def test_full_forward_losses(full_golden, full_sd, ci):
    B, N, T, L = [int(v) for v in full_golden["cases"][ci]]
    g = case_dict(full_golden, ci)
    taps = {} if ci in (0, 1) else None
    sm, st, ls = _run(synth.DEFAULT_MODEL_CFG, full_sd, B, N, T, L, 200 + ci, taps=taps)
    # 1e-3 is the north-star tolerance.  fp32 evaluation-order noise of this network is ~1e-4 at NS and
    # ~4e-4 at (T=128,N=128): the reference itself sits that far from a float64 evaluation.
    np.testing.assert_allclose(sm.numpy(), g["score_map"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(st.numpy(), g["score_tokens"], rtol=0, atol=5e-4)
    np.testing.assert_allclose([float(ls["total"]), float(ls["bce"]), float(ls["align"])], g["loss"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(ls["neg_argmax"].numpy(), g["neg_argmax"])
    np.testing.assert_array_equal(sola_oracle.select(sm).numpy(), g["selected"])
    np.testing.assert_array_equal(sm.argmax(dim=1).numpy(), g["argmax_track"])
    if taps is not None:
        for k, v in taps.items():
            if "tap." + k not in g:
                continue
            ref = g["tap." + k]
            v = v.numpy()
            if v.ndim == 4:
                v = v[:, :2]
            np.testing.assert_allclose(v, ref, rtol=0, atol=2e-4 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_float64_oracle_agrees(full_golden, full_sd):
    """The float64 evaluation of the same restatement stays within the north-star tolerance of the fp32 reference."""
    B, N, T, L = [int(v) for v in full_golden["cases"][0]]
    g = case_dict(full_golden, 0)
    sm, st, _ = _run(synth.DEFAULT_MODEL_CFG, full_sd, B, N, T, L, 200, dtype=torch.float64)
    np.testing.assert_allclose(sm.numpy(), g["score_map"], rtol=0, atol=1e-4)


def test_grad_norm_dict_restatement(small_golden):
    g = case_dict(small_golden, 0)
    grads = {k[len("grad."):]: g[k] for k in g if k.startswith("grad.")}
    out = sola_oracle.grad_norm_dict(grads, 2)
    ref = dict(zip([str(k) for k in g["grad_norm_keys"]], g["grad_norm_vals"]))
    assert set(out) == set(ref)
    for k in ref:
        assert out[k] == pytest.approx(ref[k], rel=1e-5)


# ------------------------------------------------------------------------------------------- IoU
def test_iou_pairs(iou_golden):
    A = np.unpackbits(iou_golden["pair_A"], axis=-1)[..., :960]
    B = np.unpackbits(iou_golden["pair_B"], axis=-1)[..., :960]
    ref = iou_golden["pair_iou"]
    inter, union = iou_oracle.iou_matrix(A, B)
    got = np.array([[iou_oracle.iou_from_counts(int(inter[p, r]), int(union[p, r])) for r in range(B.shape[0])]
                    for p in range(A.shape[0])])
    np.testing.assert_array_equal(got, ref)  # bit-exact doubles
    assert got[3, 5] == 1.0  # empty/empty
    np.testing.assert_array_equal(got > 0.7, ref > 0.7)


def test_iou_tie(iou_golden):
    a = np.zeros((10, 10), np.uint8)
    b = np.zeros((10, 10), np.uint8)
    a[0, :10] = 1
    b[0, :7] = 1
    v = iou_oracle.compute_mask_iou(a, b)
    assert v == float(iou_golden["tie_iou"]) == 0.7
    assert not (v > 0.7)


def test_nearest_resize_index(iou_golden):
    for i, (h, w, Ho, Wo) in enumerate(iou_golden["resize_sizes"]):
        np.testing.assert_array_equal(iou_oracle.nearest_index(int(Ho), int(h)), iou_golden[f"resize{i}_row"])
        np.testing.assert_array_equal(iou_oracle.nearest_index(int(Wo), int(w)), iou_golden[f"resize{i}_col"])


def test_masklet_iou(iou_golden):
    v = iou_oracle.compute_masklet_iou(iou_golden["masklet_A"], iou_golden["masklet_B"])
    assert v == float(iou_golden["masklet_iou"])


def test_dedup_loop(iou_golden):
    tracks = np.unpackbits(iou_golden["dedup_tracks"], axis=-1)[..., :960]
    segs = np.unpackbits(iou_golden["dedup_segs"], axis=-1)[..., :640]
    ids = [int(v) for v in iou_golden["dedup_ids"]]
    masklets = {pid: tracks[i] for i, pid in enumerate(ids)}
    prompts = [{"status": int(s), "frame_idx": int(f), "segmentation": segs[r]}
               for r, (s, f) in enumerate(zip(iou_golden["dedup_status_in"], iou_golden["dedup_frame_idx"]))]
    n = iou_oracle.dedup_batch(masklets, ids, prompts, 0.7)
    assert n == int(iou_golden["dedup_n_filtered"])
    np.testing.assert_array_equal([p["status"] for p in prompts], iou_golden["dedup_status_out"])
    np.testing.assert_array_equal([p.get("filtered_by", -1) for p in prompts], iou_golden["dedup_filtered_by"])
    np.testing.assert_array_equal([p.get("filtered_iou", -1.0) for p in prompts], iou_golden["dedup_filtered_iou"])


def test_range_cases_vs_reference():
    """tests/golden/range_golden.npz (inputs scaled by 1e-5..1e3, weights away from the default-init scale): the oracle
    follows the reference there too, at the magnitude-relative bar tests/test_gpu_range.py uses."""
    from conftest import _load

    g = _load("range_golden.npz")
    cfg = synth.DEFAULT_MODEL_CFG
    B, N, T, L = [int(v) for v in g["shape"]]
    inp = synth.make_inputs(cfg, B, N, T, L, seed=300)

    def check(sd, so, sl, g_sm, g_st, what, cond=(0.0, 0.0)):
        sm, st = sola_oracle.forward(sd, cfg, inp["object_tokens"] * np.float32(so), inp["lang_tokens"] * np.float32(sl))
        tol_sm = max(5e-4 * max(1.0, float(np.abs(g_sm).max()) / 16.0), 3.0 * float(cond[0]))
        tol_st = max(5e-4 * max(1.0, float(np.abs(g_st).max()) / 16.0), 3.0 * float(cond[1]))
        assert np.abs(sm.numpy() - g_sm).max() <= tol_sm, what
        assert np.abs(st.numpy()[:, :8] - g_st).max() <= tol_st, what

    base = sola_oracle.to_torch_state(synth.make_state_dict(cfg, 42))
    for i in (0, 2, 4, 7, 8):
        so, sl = synth.RANGE_INPUT_SCALES[i]
        check(base, so, sl, g[f"in{i}.score_map"], g[f"in{i}.score_tokens"], f"input scales {(so, sl)}")
    for v in ("lin_outliers", "gamma_div256", "gamma_x300"):
        sd = sola_oracle.to_torch_state(synth.make_state_dict_variant(cfg, 42, v))
        check(sd, 1.0, 1.0, g[f"w.{v}.score_map"], g[f"w.{v}.score_tokens"], f"weights {v}", g[f"w.{v}.cond"])


def test_accumulated_gradients_of_mixed_shape_samples_small():
    """tests/golden/ragged_train_golden.npz (gen_golden.py ragged_train): the reference stepped one sample at a time over ten
    samples of different (N, T, L) with the gradients accumulated.  The oracle, differentiated by autograd, must reproduce the
    per-sample losses and the summed gradient - it is the checker of the ragged training step on the GPU."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "ragged_train_golden.npz"), allow_pickle=False)
    cfg = synth.SMALL_MODEL_CFG
    sd = {k: torch.tensor(v, requires_grad=(k != "positional_encoding_gaussian_matrix")) for k, v in synth.make_state_dict(cfg, 42).items()}
    seed0 = int(g["small.seed0"])
    for i, (N, T, L) in enumerate(g["small.shapes"]):
        inp = synth.make_inputs(cfg, 1, int(N), int(T), int(L), seed0 + i)
        sm, st = sola_oracle.forward(sd, cfg, inp["object_tokens"], inp["lang_tokens"])
        ls = sola_oracle.losses(sm, st, inp["labels"], inp["pos_tokens"], sd["negative_token.weight"].unsqueeze(0), POS_W, TEMP, ALIGN_W)
        np.testing.assert_allclose([float(ls[k].detach()) for k in ("total", "bce", "align")], g["small.loss"][i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(sm.detach().numpy()[0], g[f"small.s{i}.score_map"], rtol=0, atol=1e-4)
        ls["total"].backward()  # accumulates
    total = float(np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g.files if k.startswith("small.gradsum."))))
    for k, t in sd.items():
        if not t.requires_grad:
            continue
        ref = g["small.gradsum." + k]
        err = float(np.abs(t.grad.numpy() - ref).max())
        assert err <= 1e-3 * float(np.abs(ref).max()) + 1e-6 * total, (k, err, float(np.abs(ref).max()))


def test_benched_batches_vs_reference(full_sd):
    """tests/golden/bench_golden.npz (gen_golden.py bench): the reference's own logits for the batches bench.py times.  The oracle on a
    bounded sample of them (CPU time): the first 8 samples of the headline batch (seed 1000), base and "lin_div64" weights, in fp32 AND
    float64 (the stored float64 row is the oracle's own: regenerated bit for bit within 1e-9), and the first 6 samples of the ragged
    training batch - one per-sample forward each, as the reference is called (inference.py:58)."""
    from conftest import _load

    g = _load("bench_golden.npz")
    cfg = synth.DEFAULT_MODEL_CFG
    inp = synth.make_inputs(cfg, 256, 64, 32, 16, seed=1000)
    tsd = sola_oracle.to_torch_state(full_sd)
    sm, _ = sola_oracle.forward(tsd, cfg, inp["object_tokens"][:8], inp["lang_tokens"][:8])
    ref = g["u256.1000.score_map"].reshape(256, 64)[:8]
    assert float(np.abs(sm.numpy() - ref).max()) <= 1e-3
    np.testing.assert_array_equal(sola_oracle.select(sm).numpy() > 0, g["u256.1000.selected"].reshape(256, 64)[:8])
    sm64, _ = sola_oracle.forward(tsd, cfg, inp["object_tokens"][:8], inp["lang_tokens"][:8], dtype=torch.float64)
    np.testing.assert_allclose(sm64.numpy(), g["u256.1000.oracle_f64.score_map"].reshape(256, 64)[:8], rtol=0, atol=1e-9)
    assert float(np.abs(sm64.numpy() - ref).max()) <= 6e-4  # the reference's own fp32 distance from exact arithmetic on these rows
    vsd = sola_oracle.to_torch_state(synth.make_state_dict_variant(cfg, 42, "lin_div64"))
    smv, _ = sola_oracle.forward(vsd, cfg, inp["object_tokens"][:8], inp["lang_tokens"][:8])
    assert float(np.abs(smv.numpy() - g["u256.1000.lin_div64.score_map"].reshape(256, 64)[:8]).max()) <= 1e-3
    smp = synth.make_ragged_samples(cfg, 64, 2024)
    counts = g["rag_train.2024.counts"]
    off = 0
    for i in range(6):
        s1, _ = sola_oracle.forward(tsd, cfg, smp[i]["obj"].numpy()[None], smp[i]["lang"].numpy()[None])
        assert int(counts[i]) == s1.shape[1]
        assert float(np.abs(s1.numpy()[0] - g["rag_train.2024.score_map"][off:off + counts[i]]).max()) <= 1e-3, i
        off += int(counts[i])
