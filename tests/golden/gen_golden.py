"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the authoring container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/gen_golden.py

The reference modules (module/module.py, tools/loss.py, track_generation/seg_utils.py) are imported
unmodified; weights and inputs come from sola_amd.synth (numpy PCG64 formulas), so only seeds and outputs
are stored.  The loss assembly mirrors train.py:98-113 with the reference's own AlignmentLoss and torch's
F.binary_cross_entropy_with_logits; the de-dup loop mirrors generate_tokens_grid.py:252-278 calling the
reference's compute_mask_iou and F.interpolate(mode="nearest").
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "track_generation"))

from sola_amd import synth  # noqa: E402

from module.module import LanguageAlignedTrackSelectionModule  # noqa: E402  (reference)
from tools.loss import AlignmentLoss  # noqa: E402  (reference)

# seg_utils imports pycocotools at module top (seg_utils.py:4); it is absent here and unused by the IoU fns.
_stub = types.ModuleType("pycocotools")
_stub.mask = types.ModuleType("pycocotools.mask")
sys.modules.setdefault("pycocotools", _stub)
sys.modules.setdefault("pycocotools.mask", _stub.mask)
import seg_utils as ref_seg  # noqa: E402  (reference)

torch.manual_seed(0)
torch.set_num_threads(8)

POS_W, TEMP, ALIGN_W = 1.5, 0.07, 0.3  # configs/mevis/default.yaml:18-27


def build_reference(cfg, seed):
    m = LanguageAlignedTrackSelectionModule(cfg)
    sd = synth.make_state_dict(cfg, seed)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval(), sd


def run_case(m, cfg, B, N, T, L, seed, want_taps, want_grads, tap_slice=None):
    inp = synth.make_inputs(cfg, B, N, T, L, seed)
    obj = torch.from_numpy(inp["object_tokens"])
    lang = torch.from_numpy(inp["lang_tokens"])
    labels = torch.from_numpy(inp["labels"])
    pos = torch.from_numpy(inp["pos_tokens"])
    taps = {}
    hooks = []
    Tp = synth.t_out_lengths(T)[-1]
    if want_taps:
        conv_idx = [0, 4, 8, 12, 16, 20]
        for li, ci in enumerate(conv_idx):
            def hk(_mod, _inp, out, li=li):
                taps[f"conv{li}"] = out.detach().reshape(B, N, out.shape[1], out.shape[2]).permute(0, 1, 3, 2).contiguous()
            hooks.append(m.short_motion_encoder[ci].register_forward_hook(hk))
        for layer_i, layer in enumerate(m.object_lang_align_layers):
            def h0(_m, _i, out, layer_i=layer_i):
                D = out.shape[1]
                taps[f"l{layer_i}_obj"] = out.detach().reshape(B, Tp, D, N).permute(0, 3, 1, 2).contiguous()
            def h1(_m, _i, out, layer_i=layer_i):
                D = out.shape[1]
                taps[f"l{layer_i}_motion"] = out.detach().reshape(B, N, D, Tp).permute(0, 1, 3, 2).contiguous()
            def h2(_m, _i, out, layer_i=layer_i):
                D = out.shape[1]
                taps[f"l{layer_i}_o2l"] = out.detach().reshape(B, D, N, Tp).permute(0, 2, 3, 1).contiguous()
            hooks += [layer.norm[0].register_forward_hook(h0), layer.norm[1].register_forward_hook(h1),
                      layer.norm[2].register_forward_hook(h2)]
    m.zero_grad(set_to_none=True)
    align_fn = AlignmentLoss(positive_weight=POS_W, temperature=TEMP)
    with torch.set_grad_enabled(bool(want_grads)):
        score_map, score_tokens = m(obj, lang)
        # train.py:92 (batch_size taken as lang_tokens.shape[0])
        neg = m.negative_token.weight.clone().unsqueeze(0).repeat(B, 1, 1)
        weight = torch.ones_like(labels)
        weight[labels > 0] = POS_W
        bce = F.binary_cross_entropy_with_logits(input=score_map, target=labels, weight=weight)
        align = align_fn(object_tokens=score_tokens, labels=labels, pos_tokens=pos, neg_tokens=neg)
        total = bce + align * ALIGN_W
        neg_logits = torch.einsum("bnd,bmd->bnm", score_tokens, neg) * torch.exp(align_fn.temperature)
        neg_argmax = neg_logits.argmax(dim=-1)
    out = {
        "score_map": score_map.detach().numpy(),
        "score_tokens": score_tokens.detach().numpy(),
        "loss": np.array([total.item(), bce.item(), align.item()], dtype=np.float64),
        "neg_argmax": neg_argmax.numpy().astype(np.int32),
        "selected": (torch.sigmoid(score_map.detach()) > 0.5).float().numpy(),  # inference.py:59-60
        "argmax_track": score_map.detach().argmax(dim=1).numpy().astype(np.int32),
    }
    if want_grads:
        total.backward()
        gnd = m.get_grad_norm_dict()
        out["grad_norm_keys"] = np.array(sorted(gnd.keys()))
        out["grad_norm_vals"] = np.array([gnd[k] for k in sorted(gnd.keys())], dtype=np.float64)
        if want_grads == "full":
            for k, p in m.named_parameters():
                out["grad." + k] = p.grad.detach().numpy().copy()
        else:
            for k, p in m.named_parameters():
                out["gradnorm." + k] = np.array(float(p.grad.detach().double().norm()))
    for h in hooks:
        h.remove()
    if want_taps:
        with torch.no_grad():
            taps["pe"] = m.get_temporal_positional_encoding(torch.zeros(1, 1, Tp, 1))[0, 0].numpy()
        for k, v in taps.items():
            v = v.numpy() if isinstance(v, torch.Tensor) else v
            if tap_slice is not None and v.ndim == 4:
                v = v[:, :tap_slice]
            out["tap." + k] = np.ascontiguousarray(v)
    return out


def gen_model_golden():
    # ---- small configuration: full taps, full grads on two cases
    cfg = synth.SMALL_MODEL_CFG
    m, _ = build_reference(cfg, seed=42)
    cases = [(1, 8, 8, 5), (2, 5, 20, 6), (1, 16, 32, 9), (2, 3, 1, 4), (1, 7, 33, 16), (1, 20, 200, 7)]
    store = {"cases": np.array(cases, dtype=np.int32)}
    for ci, (B, N, T, L) in enumerate(cases):
        r = run_case(m, cfg, B, N, T, L, seed=100 + ci, want_taps=True, want_grads="full" if ci in (0, 1) else True)
        for k, v in r.items():
            store[f"c{ci}.{k}"] = v
    np.savez_compressed(os.path.join(HERE, "small_golden.npz"), **store)
    print("small_golden.npz", os.path.getsize(os.path.join(HERE, "small_golden.npz")) / 1e6, "MB")

    # ---- full configuration (configs/mevis/default.yaml): outputs only, taps for the first 2 tracks
    cfg = synth.DEFAULT_MODEL_CFG
    m, _ = build_reference(cfg, seed=42)
    cases = [(1, 8, 8, 16), (1, 64, 32, 16), (2, 16, 32, 16), (1, 80, 32, 11), (1, 128, 128, 16)]
    store = {"cases": np.array(cases, dtype=np.int32)}
    for ci, (B, N, T, L) in enumerate(cases):
        big = N * T >= 128 * 128
        r = run_case(m, cfg, B, N, T, L, seed=200 + ci, want_taps=(ci in (0, 1)), want_grads=(not big),
                     tap_slice=2)
        for k, v in r.items():
            store[f"c{ci}.{k}"] = v
        print("full case", ci, (B, N, T, L), "loss", r["loss"], "max|score|", np.abs(r["score_map"]).max())
    np.savez_compressed(os.path.join(HERE, "full_golden.npz"), **store)
    print("full_golden.npz", os.path.getsize(os.path.join(HERE, "full_golden.npz")) / 1e6, "MB")


RAGGED_SMALL_SHAPES = [(8, 8, 5), (5, 20, 6), (16, 32, 9), (3, 1, 4), (7, 33, 16), (1, 9, 1), (20, 200, 7), (70, 17, 40), (2, 64, 3), (11, 130, 12)]
RAGGED_FULL_SHAPES = [(8, 8, 16), (64, 32, 16), (16, 32, 9), (3, 1, 4), (80, 32, 11), (7, 33, 24), (12, 150, 5)]


def gen_ragged_train_golden():
    """The reference's training step at its batch size of 1 (configs/mevis/default.yaml:37; train.py:62-117), run over samples
    of DIFFERENT (N, T, L) with the gradients ACCUMULATED (no zero_grad in between): the sum of the per-sample gradients, which
    is what one ragged step of the build (sola_forward_train_ragged / sola_backward_ragged) must produce for loss = sum of the
    per-sample totals.  Small configuration: the summed gradient in full; default configuration (33 M parameters): its
    per-parameter norms and the first 256 elements of every tensor."""
    store = {}
    for tag, cfg, shapes, seed0 in (("small", synth.SMALL_MODEL_CFG, RAGGED_SMALL_SHAPES, 300),
                                    ("full", synth.DEFAULT_MODEL_CFG, RAGGED_FULL_SHAPES, 400)):
        m, _ = build_reference(cfg, seed=42)
        m.zero_grad(set_to_none=True)
        align_fn = AlignmentLoss(positive_weight=POS_W, temperature=TEMP)
        store[f"{tag}.shapes"] = np.array(shapes, dtype=np.int32)
        store[f"{tag}.seed0"] = np.array(seed0, dtype=np.int32)
        losses = []
        for i, (N, T, L) in enumerate(shapes):
            inp = synth.make_inputs(cfg, 1, N, T, L, seed0 + i)
            obj, lang = torch.from_numpy(inp["object_tokens"]), torch.from_numpy(inp["lang_tokens"])
            labels, pos = torch.from_numpy(inp["labels"]), torch.from_numpy(inp["pos_tokens"])
            score_map, score_tokens = m(obj, lang)
            neg = m.negative_token.weight.clone().unsqueeze(0).repeat(1, 1, 1)  # train.py:92
            weight = torch.ones_like(labels)
            weight[labels > 0] = POS_W
            bce = F.binary_cross_entropy_with_logits(input=score_map, target=labels, weight=weight)
            align = align_fn(object_tokens=score_tokens, labels=labels, pos_tokens=pos, neg_tokens=neg)
            total = bce + align * ALIGN_W
            total.backward()  # accumulates into .grad
            losses.append([total.item(), bce.item(), align.item()])
            store[f"{tag}.s{i}.score_map"] = score_map.detach().numpy()[0]
            print(tag, "sample", i, (N, T, L), "loss", losses[-1])
        store[f"{tag}.loss"] = np.array(losses, dtype=np.float64)
        for k, p in m.named_parameters():
            g = p.grad.detach()
            if tag == "small":
                store[f"{tag}.gradsum.{k}"] = g.numpy().copy()
            else:
                store[f"{tag}.gradsum_norm.{k}"] = np.array(float(g.double().norm()))
                store[f"{tag}.gradsum_head.{k}"] = g.reshape(-1)[:256].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "ragged_train_golden.npz"), **store)
    print("ragged_train_golden.npz", os.path.getsize(os.path.join(HERE, "ragged_train_golden.npz")) / 1e6, "MB")


def gen_range_golden():
    """Reference outputs for inputs away from N(0,1) and weights away from the default-init scale (the range cases of the
    split-f16 mode): full configuration at the headline shape, outputs only."""
    cfg = synth.DEFAULT_MODEL_CFG
    B, N, T, L = 1, 64, 32, 16
    store = {"shape": np.array([B, N, T, L], dtype=np.int32), "input_scales": np.array(synth.RANGE_INPUT_SCALES, dtype=np.float64),
             "weight_variants": np.array(synth.WEIGHT_VARIANTS)}
    inp = synth.make_inputs(cfg, B, N, T, L, seed=300)

    from oracle import sola_oracle  # float64 evaluation: how far the reference's own fp32 result is from exact arithmetic

    def run(m, so, sl):
        with torch.no_grad():
            sm, st = m(torch.from_numpy(inp["object_tokens"] * np.float32(so)), torch.from_numpy(inp["lang_tokens"] * np.float32(sl)))
        return sm.numpy(), st.numpy()

    def cond(sd, so, sl, sm, st):
        osm, ost = sola_oracle.forward(sola_oracle.to_torch_state(sd), cfg, inp["object_tokens"] * np.float32(so),
                                       inp["lang_tokens"] * np.float32(sl), dtype=torch.float64)
        return np.array([np.abs(osm.numpy() - sm).max(), np.abs(ost.numpy() - st).max()])

    m, base_sd = build_reference(cfg, seed=42)
    for i, (so, sl) in enumerate(synth.RANGE_INPUT_SCALES):
        sm, st = run(m, so, sl)
        store[f"in{i}.score_map"], store[f"in{i}.score_tokens"] = sm, st[:, :8]  # tokens of the first 8 tracks
        store[f"in{i}.cond"] = cond(base_sd, so, sl, sm, st)
        print("input scales", (so, sl), "max|score|", np.abs(sm).max(), "max|tokens|", np.abs(st).max(), "cond", store[f"in{i}.cond"])
    for v in synth.WEIGHT_VARIANTS:
        m = LanguageAlignedTrackSelectionModule(cfg)
        sd = synth.make_state_dict_variant(cfg, 42, v)
        m.load_state_dict({k: torch.from_numpy(a) for k, a in sd.items()}, strict=True)
        sm, st = run(m.eval(), 1.0, 1.0)
        store[f"w.{v}.score_map"], store[f"w.{v}.score_tokens"] = sm, st[:, :8]
        store[f"w.{v}.cond"] = cond(sd, 1.0, 1.0, sm, st)
        print("weights", v, "max|score|", np.abs(sm).max(), "max|tokens|", np.abs(st).max(), "cond", store[f"w.{v}.cond"])
    np.savez_compressed(os.path.join(HERE, "range_golden.npz"), **store)
    print("range_golden.npz", os.path.getsize(os.path.join(HERE, "range_golden.npz")) / 1e6, "MB")


def gen_bench_golden():
    """The reference's OWN logits for every batch bench.py times and the every-row GPU tests check (those batches
    were compared with the oracle only).  The reference is called the way its entry points call it - ONE sample per forward
    (inference.py:58, evaluator.py:100, train.py:95 at configs/mevis/default.yaml:37), eval mode - and only score_map is stored
    (float32; the selection sigmoid(score_map) > 0.5 of inference.py:59-60 as a bool array beside it):
      u256.<seed>              256 samples at (T=32, N=64, L=16), seeds 1000 (bench.py's rank-0 batch), 1001, 1002
      u256.1000.oracle_f64     the same samples through oracle/sola_oracle.py in float64 (the reference itself cannot run in double:
                               module/module.py:120-125 builds its positional table in float32): how far the reference's own
                               fp32 result sits from exact arithmetic, row by row
      u256.1000.lin_div64      seed 1000 on weights whose first softmax is NOT saturated (synth variant "lin_div64")
      c4.<seed>                BASELINE config C4 (T=128, N=128, L=16), 32 samples: seed 2000 (tests), 77 (bench.py's stress leg)
      rag_infer.<tag>          synth.make_ragged_infer_batches(128, 2024): one / four expressions per video, logits concatenated
      rag_train.<seed>         synth.make_ragged_samples(64, 2024) / (128, 2025): the ragged training batches (eval-mode forward)
    """
    cfg = synth.DEFAULT_MODEL_CFG
    store = {}
    m, _ = build_reference(cfg, seed=42)

    def per_sample(mod, objs, langs, dtype=torch.float32):
        out = []
        with torch.no_grad():
            for o, l in zip(objs, langs):
                sm, _st = mod(torch.from_numpy(o)[None].to(dtype), torch.from_numpy(l)[None].to(dtype))
                out.append(sm[0].numpy())
        return out

    def put(key, rows):
        flat = np.concatenate(rows)
        store[key + ".score_map"] = flat.astype(np.float32) if flat.dtype == np.float32 else flat
        store[key + ".selected"] = (torch.sigmoid(torch.from_numpy(flat)) > 0.5).numpy()  # inference.py:59-60
        print(key, flat.shape, "max|logit|", float(np.abs(flat).max()), flush=True)

    B, N, T, L = 256, 64, 32, 16
    for seed in (1000, 1001, 1002):
        inp = synth.make_inputs(cfg, B, N, T, L, seed=seed)
        put(f"u256.{seed}", per_sample(m, inp["object_tokens"], inp["lang_tokens"]))
    inp = synth.make_inputs(cfg, B, N, T, L, seed=1000)
    from oracle import sola_oracle
    tsd = sola_oracle.to_torch_state(synth.make_state_dict(cfg, 42))
    rows64 = [sola_oracle.forward(tsd, cfg, inp["object_tokens"][b:b + 8], inp["lang_tokens"][b:b + 8], dtype=torch.float64)[0].numpy().reshape(-1)
              for b in range(0, B, 8)]
    put("u256.1000.oracle_f64", rows64)
    mv = LanguageAlignedTrackSelectionModule(cfg)
    mv.load_state_dict({k: torch.from_numpy(a) for k, a in synth.make_state_dict_variant(cfg, 42, "lin_div64").items()}, strict=True)
    put("u256.1000.lin_div64", per_sample(mv.eval(), inp["object_tokens"], inp["lang_tokens"]))
    del mv
    for seed in (2000, 77):
        inp = synth.make_inputs(cfg, 32, 128, 128, 16, seed=seed)
        put(f"c4.{seed}", per_sample(m, inp["object_tokens"], inp["lang_tokens"]))
    for tag, bt in synth.make_ragged_infer_batches(cfg, 128, 2024).items():
        objs = [bt["videos"][v] for v in bt["sample_video"]]
        put(f"rag_infer.{tag}", per_sample(m, objs, bt["texts"]))
        store[f"rag_infer.{tag}.counts"] = np.array([o.shape[0] for o in objs], dtype=np.int32)
    for n, seed in ((64, 2024), (128, 2025)):
        smp = synth.make_ragged_samples(cfg, n, seed)
        put(f"rag_train.{seed}", per_sample(m, [x["obj"].numpy() for x in smp], [x["lang"].numpy() for x in smp]))
        store[f"rag_train.{seed}.counts"] = np.array([int(x["obj"].shape[0]) for x in smp], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "bench_golden.npz"), **store)
    print("bench_golden.npz", os.path.getsize(os.path.join(HERE, "bench_golden.npz")) / 1e6, "MB")


def rect_masks(rng, n, H, W, base=None, jitter=0):
    out = np.zeros((n, H, W), dtype=np.uint8)
    boxes = []
    for i in range(n):
        if base is None:
            y0, x0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
            y1, x1 = y0 + rng.integers(4, H // 2), x0 + rng.integers(4, W // 2)
        else:
            by0, bx0, by1, bx1 = base[i % len(base)]
            y0 = int(np.clip(by0 + rng.integers(-jitter, jitter + 1), 0, H - 2))
            x0 = int(np.clip(bx0 + rng.integers(-jitter, jitter + 1), 0, W - 2))
            y1 = int(np.clip(by1 + rng.integers(-jitter, jitter + 1), y0 + 1, H))
            x1 = int(np.clip(bx1 + rng.integers(-jitter, jitter + 1), x0 + 1, W))
        out[i, y0:y1, x0:x1] = 1
        boxes.append((y0, x0, y1, x1))
    return out, boxes


def gen_iou_golden():
    rng = np.random.Generator(np.random.PCG64(7))
    store = {}
    # (1) pairwise IoU at the comparison resolution, IoUs clustered around 0.7, via reference compute_mask_iou
    H, W = 540, 960
    A, boxes = rect_masks(rng, 4, H, W)
    B, _ = rect_masks(rng, 24, H, W, base=boxes, jitter=40)
    B[5] = 0  # empty prompt
    A2 = A.copy()
    A2[3] = 0  # empty track: union==0 case against B[5]
    iou = np.zeros((4, 24), dtype=np.float64)
    for p in range(4):
        for r in range(24):
            iou[p, r] = ref_seg.compute_mask_iou(torch.from_numpy(A2[p]).float(), torch.from_numpy(B[r]).float())
    store["pair_seed"] = np.array(7)
    store["pair_A"] = np.packbits(A2, axis=-1)
    store["pair_B"] = np.packbits(B, axis=-1)
    store["pair_iou"] = iou
    # exact ties: 7/10 must not pass "> 0.7"
    ta = np.zeros((1, 10, 10), np.uint8)
    tb = np.zeros((1, 10, 10), np.uint8)
    ta[0, 0, :10] = 1            # 10 px
    tb[0, 0, :7] = 1             # 7 px inside -> inter 7, union 10
    store["tie_iou"] = np.array(ref_seg.compute_mask_iou(torch.from_numpy(ta[0]).float(), torch.from_numpy(tb[0]).float()))
    # (2) nearest resize index maps for non-integer scales, via F.interpolate on an index image
    sizes = [(720, 1280, 540, 960), (1080, 1920, 540, 960), (480, 854, 540, 960), (360, 640, 540, 960),
             (1280, 720, 960, 540), (100, 37, 540, 960), (540, 960, 540, 960), (270, 480, 540, 960)]
    store["resize_sizes"] = np.array(sizes, dtype=np.int32)
    for i, (h, w, Ho, Wo) in enumerate(sizes):
        idx = torch.arange(h * w, dtype=torch.float32).reshape(1, 1, h, w)
        out = F.interpolate(idx, size=(Ho, Wo), mode="nearest")[0, 0].long()
        store[f"resize{i}_row"] = (out[:, 0] // w).numpy().astype(np.int32)
        store[f"resize{i}_col"] = (out[0, :] % w).numpy().astype(np.int32)
    # (3) masklet IoU
    ma, _ = rect_masks(rng, 6, 64, 96)
    mb, _ = rect_masks(rng, 6, 64, 96)
    store["masklet_A"], store["masklet_B"] = ma, mb
    store["masklet_iou"] = np.array(ref_seg.compute_masklet_iou(torch.from_numpy(ma).float(), torch.from_numpy(mb).float(), "cpu"))
    # (4) greedy de-dup loop (generate_tokens_grid.py:252-278) on a synthetic prompt set
    T, h, w = 3, 360, 640
    tracks, tboxes = rect_masks(rng, 3, H, W)
    masklets = {pid: np.stack([tracks[i]] * T) for i, pid in enumerate([11, 4, 7])}
    for pid in masklets:  # make frames differ
        masklets[pid][1] = np.roll(masklets[pid][1], 5, axis=1)
        masklets[pid][2] = np.roll(masklets[pid][2], -9, axis=0)
    scaled = [(int(y0 * h / H), int(x0 * w / W), int(y1 * h / H), int(x1 * w / W)) for (y0, x0, y1, x1) in tboxes]
    segs, _ = rect_masks(rng, 20, h, w, base=scaled, jitter=25)
    prompts = []
    for r in range(20):
        prompts.append({"status": 1 if r in (2, 9) else 0, "frame_idx": int(rng.integers(0, T)), "segmentation": segs[r]})
    n_filtered = 0
    for pid in [11, 4, 7]:
        for info in prompts:
            if info["status"] > 0:
                continue
            pred_mask = torch.from_numpy(masklets[pid][info["frame_idx"]]).float()
            hh, ww = pred_mask.shape
            pm = torch.from_numpy(info["segmentation"]).float()
            pm = F.interpolate(pm.unsqueeze(0).unsqueeze(0), size=(hh, ww), mode="nearest").squeeze(0).squeeze(0)
            v = ref_seg.compute_mask_iou(pred_mask, pm)
            if v > 0.7:
                info["status"] = 2
                info["filtered_by"] = pid
                info["filtered_iou"] = v
                n_filtered += 1
    store["dedup_tracks"] = np.packbits(np.stack([masklets[p] for p in [11, 4, 7]]), axis=-1)
    store["dedup_ids"] = np.array([11, 4, 7], dtype=np.int32)
    store["dedup_segs"] = np.packbits(segs, axis=-1)
    store["dedup_frame_idx"] = np.array([p["frame_idx"] for p in prompts], dtype=np.int32)
    store["dedup_status_in"] = np.array([1 if r in (2, 9) else 0 for r in range(20)], dtype=np.int32)
    store["dedup_status_out"] = np.array([p["status"] for p in prompts], dtype=np.int32)
    store["dedup_filtered_by"] = np.array([p.get("filtered_by", -1) for p in prompts], dtype=np.int32)
    store["dedup_filtered_iou"] = np.array([p.get("filtered_iou", -1.0) for p in prompts], dtype=np.float64)
    store["dedup_n_filtered"] = np.array(n_filtered)
    print("dedup filtered", n_filtered, "of 18; pair iou >0.7:", int((iou > 0.7).sum()), "of", iou.size)
    np.savez_compressed(os.path.join(HERE, "iou_golden.npz"), **store)
    print("iou_golden.npz", os.path.getsize(os.path.join(HERE, "iou_golden.npz")) / 1e6, "MB")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "iou"):
        gen_iou_golden()
    if which in ("all", "model"):
        gen_model_golden()
    if which in ("all", "range"):
        gen_range_golden()
    if which in ("all", "ragged_train"):
        gen_ragged_train_golden()
    if which in ("all", "bench"):
        gen_bench_golden()
