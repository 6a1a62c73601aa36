"""Ragged batches (sola_forward_ragged / sola_loss_ragged): many (video, expression) samples of DIFFERENT (N, T, L) in one
pass, with the text-independent part of the network computed once per video.

* every sample of a ragged batch equals its own one-sample call (the reference's batch size of 1:
  configs/mevis/default.yaml:37,42,47; inference.py:44-58) and the reference's golden vectors;
* several expressions per video share the encoder + layer-0 object/motion sub-blocks and still equal per-sample calls;
* shapes: T' = 1, odd lengths, one track, more than 64 tracks (split-f16 attention shape), more than 16 encoded steps
  (motion attention leaves the packed 16-step shape), text lengths 1..40."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import _load  # noqa: E402

from conftest import case_dict  # noqa: E402
from sola_amd import SolaError, synth  # noqa: E402
from sola_amd.loss import track_selection_losses, track_selection_losses_ragged  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

POS_W, TEMP, ALIGN_W = 1.5, 0.07, 0.3


def build(cfg, precision):
    m = LanguageAlignedTrackSelectionModule(cfg)
    sd = synth.make_state_dict(cfg, 42)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    m = m.cuda().eval()
    m.precision = precision
    return m


@pytest.fixture(scope="module", params=["f32", "f16x3"])
def small(request):
    return build(synth.SMALL_MODEL_CFG, request.param)


@pytest.fixture(scope="module", params=["f32", "f16x3"])
def full(request):
    return build(synth.DEFAULT_MODEL_CFG, request.param)


def make_videos(cfg, shapes, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [torch.from_numpy(rng.standard_normal((N, T, cfg["object_token_dim"])).astype(np.float32)).cuda() for N, T in shapes]


def make_texts(cfg, lens, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [torch.from_numpy(rng.standard_normal((L, cfg["lang_token_dim"])).astype(np.float32)).cuda() for L in lens]


def per_sample(m, videos, texts, sample_video):
    outs = []
    for i, v in enumerate(sample_video):
        with torch.no_grad():
            sm, st = m(videos[v][None], texts[i][None])
        outs.append((sm[0].clone(), st[0].clone()))
    return outs


SHAPES = [(8, 8), (5, 20), (16, 32), (3, 1), (7, 33), (1, 9), (20, 200), (70, 17), (2, 64), (11, 130)]
LENS = [5, 6, 9, 4, 16, 1, 7, 40, 3, 12]


def test_ragged_equals_per_sample_small(small):
    cfg = synth.SMALL_MODEL_CFG
    videos, texts = make_videos(cfg, SHAPES, 1), make_texts(cfg, LENS, 2)
    sms, sts = small.forward_ragged(videos, texts)
    ref = per_sample(small, videos, texts, list(range(len(videos))))
    for i, ((rsm, rst), sm, st) in enumerate(zip(ref, sms, sts)):
        assert sm.shape == rsm.shape and st.shape == rst.shape
        torch.testing.assert_close(sm, rsm, rtol=0, atol=2e-4, msg=lambda s, i=i: f"sample {i} {SHAPES[i]}: {s}")
        torch.testing.assert_close(st, rst, rtol=0, atol=2e-4, msg=lambda s, i=i: f"sample {i} {SHAPES[i]}: {s}")


def test_shared_videos_equal_per_sample_small(small):
    """14 expressions over 4 videos (one video unused by any sample would be legal too): the per-video half is computed once."""
    cfg = synth.SMALL_MODEL_CFG
    shapes = [(6, 24), (17, 40), (3, 9), (66, 16)]
    sample_video = [0, 1, 1, 3, 0, 2, 1, 3, 3, 0, 2, 2, 1, 0]
    lens = [3, 8, 1, 20, 5, 6, 7, 2, 33, 4, 9, 10, 11, 12]
    videos, texts = make_videos(cfg, shapes, 3), make_texts(cfg, lens, 4)
    sms, sts = small.forward_ragged(videos, texts, sample_video)
    ref = per_sample(small, videos, texts, sample_video)
    for i, ((rsm, rst), sm, st) in enumerate(zip(ref, sms, sts)):
        torch.testing.assert_close(sm, rsm, rtol=0, atol=2e-4, msg=lambda s, i=i: f"sample {i}: {s}")
        torch.testing.assert_close(st, rst, rtol=0, atol=2e-4, msg=lambda s, i=i: f"sample {i}: {s}")


def test_all_golden_cases_in_one_ragged_batch(full, full_golden):
    """The reference's golden samples of five different shapes - incl. (N=128, T=128) and (N=80) - scored in ONE call."""
    cfg = synth.DEFAULT_MODEL_CFG
    videos, texts, gold = [], [], []
    for ci in range(5):
        B, N, T, L = [int(v) for v in full_golden["cases"][ci]]
        inp = synth.make_inputs(cfg, B, N, T, L, 200 + ci)
        g = case_dict(full_golden, ci)
        for b in range(B):
            videos.append(torch.from_numpy(inp["object_tokens"][b]).cuda())
            texts.append(torch.from_numpy(inp["lang_tokens"][b]).cuda())
            gold.append((g["score_map"][b], g["score_tokens"][b], g["selected"][b], int(g["argmax_track"][b])))
    sms, sts = full.forward_ragged(videos, texts)
    assert full.split_fallbacks()[1] == 0
    for i, (gsm, gst, gsel, garg) in enumerate(gold):
        sm, st = sms[i].cpu().numpy(), sts[i].cpu().numpy()
        assert np.abs(sm - gsm).max() <= 1e-3 and np.abs(st - gst).max() <= 1e-3, (i, np.abs(sm - gsm).max(), np.abs(st - gst).max())
        np.testing.assert_array_equal((1.0 / (1.0 + np.exp(-sm)) > 0.5).astype(np.float32), gsel)
        assert int(sm.argmax()) == garg


def test_golden_ragged_batch_with_the_register_and_resident_attention_shapes(full, full_golden):
    """Default routing keeps ragged batches on the high-occupancy LDS shape (measured faster on mixed unit sizes); the
    resident-K/V (attn_res.hip) and register-only (attn_reg.hip) shapes read the same unit tables - forced here."""
    from sola_amd import _lib
    try:
        _lib.check(_lib.lib().sola_tune(b"attn_res", 2), "tune")
        _lib.check(_lib.lib().sola_tune(b"attn_reg", 2), "tune")
        test_all_golden_cases_in_one_ragged_batch(full, full_golden)
    finally:
        _lib.check(_lib.lib().sola_tune(b"attn_res", 1), "tune")
        _lib.check(_lib.lib().sola_tune(b"attn_reg", 1), "tune")


def test_small_golden_cases_in_one_ragged_batch(small, small_golden):
    cfg = synth.SMALL_MODEL_CFG
    videos, texts, gold = [], [], []
    for ci in range(6):
        B, N, T, L = [int(v) for v in small_golden["cases"][ci]]
        inp = synth.make_inputs(cfg, B, N, T, L, 100 + ci)
        g = case_dict(small_golden, ci)
        for b in range(B):
            videos.append(torch.from_numpy(inp["object_tokens"][b]).cuda())
            texts.append(torch.from_numpy(inp["lang_tokens"][b]).cuda())
            gold.append((g["score_map"][b], g["score_tokens"][b]))
    sms, sts = small.forward_ragged(videos, texts)
    for i, (gsm, gst) in enumerate(gold):
        assert np.abs(sms[i].cpu().numpy() - gsm).max() <= 1e-3 and np.abs(sts[i].cpu().numpy() - gst).max() <= 1e-3, i


def test_ragged_losses_equal_per_sample_losses(full):
    cfg = synth.DEFAULT_MODEL_CFG
    shapes, lens = [(16, 32), (64, 32), (9, 40), (80, 24)], [16, 7, 11, 3]
    sample_video = [0, 1, 1, 2, 3, 0]
    lens = lens + [5, 21]
    videos, texts = make_videos(cfg, shapes, 5), make_texts(cfg, lens, 6)
    rng = np.random.Generator(np.random.PCG64(7))
    labels = [torch.from_numpy((rng.uniform(size=shapes[v][0]) < 0.3).astype(np.float32)).cuda() for v in sample_video]
    pos = torch.stack([t.mean(0) for t in texts], 0)
    sms, sts = full.forward_ragged(videos, texts, sample_video)
    flat_sm, flat_st, offs, counts = full.last_ragged
    neg = full.negative_token.weight.detach()  # return_argmax is an evaluation option: no graph
    loss, argmax = track_selection_losses_ragged(flat_sm, flat_st, torch.cat(labels), pos, neg, offs, counts, POS_W, TEMP, ALIGN_W,
                                                 return_argmax=True)
    with pytest.raises(SolaError):  # ... and asking for it on the differentiable path is an error, not a loss without a graph
        track_selection_losses_ragged(flat_sm, flat_st, torch.cat(labels), pos, full.negative_token.weight, offs, counts, POS_W, TEMP,
                                      ALIGN_W, return_argmax=True)
    assert loss.shape == (len(sample_video), 3)
    o = 0
    for i in range(len(sample_video)):
        l3, am = track_selection_losses(sms[i][None], sts[i][None], labels[i][None], pos[i][None, None], neg, POS_W, TEMP, ALIGN_W,
                                        return_argmax=True)
        torch.testing.assert_close(loss[i], l3, rtol=1e-5, atol=1e-6)
        assert torch.equal(argmax[o:o + counts[i]], am[0])
        o += counts[i]


def test_ragged_argument_errors(small):
    cfg = synth.SMALL_MODEL_CFG
    videos, texts = make_videos(cfg, [(4, 8), (3, 5)], 8), make_texts(cfg, [3, 4], 9)
    with pytest.raises(SolaError):
        small.forward_ragged(videos, texts[:1])  # sample count != video count without sample_video
    with pytest.raises(SolaError):
        small.forward_ragged(videos, texts, [0, 2])  # video index out of range
    with pytest.raises(SolaError):
        small.forward_ragged([videos[0][:, :, :8]], texts[:1])  # wrong token width
    with pytest.raises(SolaError):
        small.forward_ragged([v.cpu() for v in videos], texts)  # no CPU path
    sms, _ = small.forward_ragged(videos, texts)  # still usable afterwards
    assert [tuple(s.shape) for s in sms] == [(4,), (3,)]


def test_ragged_is_deterministic_and_independent_of_batch_composition(full):
    cfg = synth.DEFAULT_MODEL_CFG
    shapes, lens = [(16, 32), (64, 32), (9, 40)], [16, 7, 11]
    videos, texts = make_videos(cfg, shapes, 10), make_texts(cfg, lens, 11)
    a, _ = full.forward_ragged(videos, texts)
    b, _ = full.forward_ragged(videos, texts)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # the same samples in another order and next to other samples: same values up to summation order
    c, _ = full.forward_ragged([videos[2], videos[0], videos[1], videos[0]], [texts[2], texts[0], texts[1], texts[0]])
    torch.testing.assert_close(c[1], a[0], rtol=0, atol=2e-4)
    torch.testing.assert_close(c[2], a[1], rtol=0, atol=2e-4)
    torch.testing.assert_close(c[0], a[2], rtol=0, atol=2e-4)
    assert torch.equal(c[1], c[3])


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15])
def test_random_ragged_batches_equal_per_sample(small, seed):
    """Seeded random batches: 1..12 videos of N in [1,40] tracks x T in [1,70] frames, S in [V,3V] samples assigned to random
    videos (some videos serve several expressions, some none), L in [1,20]: every sample equals its own one-sample call."""
    cfg = synth.SMALL_MODEL_CFG
    rng = np.random.Generator(np.random.PCG64(seed))
    V = int(rng.integers(1, 13))
    shapes = [(int(rng.integers(1, 41)), int(rng.integers(1, 71))) for _ in range(V)]
    S = int(rng.integers(V, 3 * V + 1))
    sample_video = [int(v) for v in rng.integers(0, V, size=S)]
    lens = [int(v) for v in rng.integers(1, 21, size=S)]
    videos, texts = make_videos(cfg, shapes, seed * 7), make_texts(cfg, lens, seed * 7 + 1)
    sms, sts = small.forward_ragged(videos, texts, sample_video)
    ref = per_sample(small, videos, texts, sample_video)
    for i, ((rsm, rst), sm, st) in enumerate(zip(ref, sms, sts)):
        torch.testing.assert_close(sm, rsm, rtol=0, atol=2e-4, msg=lambda s, i=i: f"seed {seed} sample {i} video {shapes[sample_video[i]]} L {lens[i]}: {s}")
        torch.testing.assert_close(st, rst, rtol=0, atol=2e-4, msg=lambda s, i=i: f"seed {seed} sample {i}: {s}")


def test_ragged_under_the_16_bit_storage_mode():
    """precision "f16" on the ragged path (round 3; inference.py / eval.py only call this path): the 16-bit storage arithmetic of
    forward_f16.hip on concatenated rows - f16 activations, one f16 MFMA per product, the f16 attention kernel on unit tables.
    A reduced-precision mode with the stated tolerance of tests/test_gpu_f16.py against the exact-f32 ragged call: logits within
    TOL_LOGIT everywhere and 1.5 % rms, decisions equal away from the threshold; served by the f16 kernels (no guard repeat)."""
    from test_gpu_f16 import TOL_LOGIT, TOL_RMS
    cfg = synth.DEFAULT_MODEL_CFG
    m16, m32 = build(cfg, "f16"), build(cfg, "f32")
    shapes = [(9, 24), (20, 40), (64, 32), (3, 1), (70, 17), (12, 150)]
    sample_video = [0, 1, 1, 2, 3, 4, 5, 5, 2]
    videos, texts = make_videos(cfg, shapes, 21), make_texts(cfg, [5, 8, 3, 16, 4, 40, 7, 1, 12], 22)
    a, at = m16.forward_ragged(videos, texts, sample_video)
    assert m16.split_fallbacks() == (0, 0)
    b, bt = m32.forward_ragged(videos, texts, sample_video)
    fa, fb = torch.cat(a), torch.cat(b)
    assert float((fa - fb).abs().max()) <= TOL_LOGIT, float((fa - fb).abs().max())
    assert float((torch.cat(at) - torch.cat(bt)).abs().max()) <= TOL_LOGIT
    rms = lambda x: float(x.double().pow(2).mean().sqrt())
    assert rms(fa - fb) <= TOL_RMS * rms(fb), (rms(fa - fb), rms(fb))
    clear = fb.abs() > 2 * TOL_LOGIT
    assert torch.equal((fa > 0)[clear], (fb > 0)[clear])
    # a value beyond the f16 range trips the guard: the ragged call is repeated on the exact-f32 kernels, bit for bit
    with torch.no_grad():
        m16.short_motion_encoder[20].bias[3] = 1.0e5
        m32.short_motion_encoder[20].bias[3] = 1.0e5
    a2, _ = m16.forward_ragged(videos, texts, sample_video)
    b2, _ = m32.forward_ragged(videos, texts, sample_video)
    n, bits = m16.split_fallbacks()
    assert n == 1 and (bits & 1)
    for x, y in zip(a2, b2):
        assert torch.equal(x, y)


# ---- every logit of the benched ragged batches against the fp32 oracle --------------------------------------------
_ORACLE_ROWS = {}


def _oracle_rows(tag, bt):
    """fp32 oracle logits of every sample of a ragged batch (one per-sample forward each), cached per batch across the two precisions."""
    if tag not in _ORACLE_ROWS:
        from oracle import sola_oracle  # checker only

        torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 32))
        tsd = sola_oracle.to_torch_state(synth.make_state_dict(synth.DEFAULT_MODEL_CFG, 42))
        rows = []
        for i, v in enumerate(bt["sample_video"]):
            sm, _ = sola_oracle.forward(tsd, synth.DEFAULT_MODEL_CFG, bt["videos"][v][None], bt["texts"][i][None])
            rows.append(np.asarray(sm)[0])
        _ORACLE_ROWS[tag] = np.concatenate(rows)
    return _ORACLE_ROWS[tag]


@pytest.mark.parametrize("tag", ["one_expression_per_video", "four_expressions_per_video"])
def test_benched_ragged_batch_every_logit_vs_oracle(full, tag):
    """bench.py's ragged leg: 128 samples of the MeViS-like mix in ONE sola_forward_ragged call, (a) one expression per video, (b) four
    expressions per video (the text-independent half computed once per video).  EVERY logit of the batch against the fp32 oracle's
    per-sample forward: the north star's 1e-3, selections equal - in exact f32 and in the default split-f16 mode (fixture param)."""
    bt = synth.make_ragged_infer_batches(synth.DEFAULT_MODEL_CFG, 128, 2024)[tag]
    ref = _oracle_rows(tag, bt)
    videos = [torch.from_numpy(v).cuda() for v in bt["videos"]]
    texts = [torch.from_numpy(t).cuda() for t in bt["texts"]]
    full.forward_ragged(videos, texts, bt["sample_video"])
    flat, _tok, _offs, counts = full.last_ragged
    got = flat.cpu().numpy()
    assert got.shape == ref.shape and sum(counts) == got.shape[0]
    e = np.abs(got - ref)
    starts = np.cumsum([0] + list(counts[:-1]))
    per = np.array([e[o:o + c].max() for o, c in zip(starts, counts)])
    print(f"{tag} {full.precision}: worst logit error {e.max():.3e}, mean per-sample worst {per.mean():.3e}, samples > 5e-4: {(per > 5e-4).sum()} of {len(counts)}")
    np.testing.assert_array_equal(got > 0, ref > 0)
    assert e.max() <= 1e-3
    if full.precision == "f16x3":
        assert full.split_fallbacks()[1] == 0  # the range guard did not trip: these are the split-f16 kernels' numbers
    # round 6: the same logits against the REFERENCE's own (one module/module.py forward per sample; tests/golden/gen_golden.py bench)
    gold = _load("bench_golden.npz")
    rref = gold[f"rag_infer.{tag}.score_map"]
    assert list(gold[f"rag_infer.{tag}.counts"]) == list(counts)
    er = np.abs(got - rref)
    perr = np.array([er[o:o + c].max() for o, c in zip(starts, counts)])
    print(f"{tag} {full.precision} vs REFERENCE: worst logit error {er.max():.3e}, mean per-sample worst {perr.mean():.3e}, samples > 5e-4: {(perr > 5e-4).sum()}")
    np.testing.assert_array_equal((torch.sigmoid(torch.from_numpy(got)) > 0.5).numpy(), gold[f"rag_infer.{tag}.selected"])
    assert er.max() <= 1e-3


@pytest.mark.gpu
def test_collated_batch_gives_the_same_ragged_forward_bit_for_bit():
    """A batch handed over as views of one buffer (module.collate_ragged) is read where it lies - the same call on the same values as the
    list of separate tensors that forward_ragged concatenates."""
    import torch
    from sola_amd import synth
    from sola_amd.module import LanguageAlignedTrackSelectionModule, collate_ragged

    cfg = synth.DEFAULT_MODEL_CFG
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()}, strict=True)
    m = m.cuda().eval()
    smp = synth.make_ragged_samples(cfg, 6, 77, "cuda", n_range=(3, 12), t_range=(9, 40))
    objs, langs = [x["obj"] for x in smp], [x["lang"] for x in smp]
    with torch.no_grad():
        a_map, a_tok = m.forward_ragged(objs, langs)
        b_map, b_tok = m.forward_ragged(collate_ragged(objs), collate_ragged(langs))
    assert all(torch.equal(x, y) for x, y in zip(a_map, b_map)) and all(torch.equal(x, y) for x, y in zip(a_tok, b_tok))
