"""Drop-in for the IoU functions of ``track_generation/seg_utils.py`` (:109-142) and the greedy de-dup loop of
``generate_tokens_grid.py:252-278`` / ``generate_tokens_gdino.py:274-300`` on libsola_hip.so.

Masks are torch CUDA tensors, uint8 or float32 with values {0,1}.  Counts are exact int64; the ratio is a python
float division exactly as in the reference, so ``iou > miou_thresh`` decisions are bit-identical."""
from __future__ import annotations

import torch

from ._lib import SolaError, check, current_stream, lib, ptr, require_cuda


def _elem_type(t):
    if t.dtype == torch.uint8 or t.dtype == torch.bool:
        return 0
    if t.dtype == torch.float32:
        return 1
    raise SolaError(f"masks must be uint8/bool or float32, got {t.dtype}")


def _prep(t):
    require_cuda(t)
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    return t.contiguous()


def pack_masks(masks, out_hw=None):
    """masks [n,h,w] -> (bits int32 [n,words], area int64 [n]) at resolution ``out_hw`` (nearest resample, as
    F.interpolate(mode='nearest') in generate_tokens_grid.py:272) or the native one."""
    masks = _prep(masks)
    n, h, w = masks.shape
    H, W = (h, w) if out_hw is None else out_hw
    words = lib().sola_mask_words(H, W)
    bits = torch.empty((n, words), device=masks.device, dtype=torch.int32)
    area = torch.empty((n,), device=masks.device, dtype=torch.int64)
    check(lib().sola_mask_pack(ptr(masks), _elem_type(masks), n, h, w, H, W, ptr(bits), ptr(area),
                               current_stream(masks.device)), "sola_mask_pack")
    return bits, area


def pair_counts(a_bits, a_area, b_bits, b_area, T=1, a_frame=None):
    """inter/union [P,R] from packed masks; ``a_frame`` [R] int32 selects, per column, the frame of each of the P
    masklets of T frames held in ``a_bits`` [P*T, words]."""
    P = a_bits.shape[0] // T
    R = b_bits.shape[0]
    words = a_bits.shape[1]
    if b_bits.shape[1] != words:
        raise SolaError("packed masks have different resolutions")
    dev = a_bits.device
    inter = torch.empty((P, R), device=dev, dtype=torch.int64)
    union = torch.empty((P, R), device=dev, dtype=torch.int64)
    if a_frame is not None:
        a_frame = a_frame.to(device=dev, dtype=torch.int32).contiguous()
    check(lib().sola_mask_pair_counts(ptr(a_bits), ptr(a_area), P, T, ptr(b_bits), ptr(b_area), R, ptr(a_frame), words,
                                      ptr(inter), ptr(union), current_stream(dev)), "sola_mask_pair_counts")
    return inter, union


_IOU_SCRATCH = {}


def mask_iou_matrix(A, B):
    """A [P,H,W], B [R,h,w] (resampled to H x W) -> (inter, union) int64 [P,R] in one library call."""
    A, B = _prep(A), _prep(B)
    if _elem_type(A) != _elem_type(B):
        raise SolaError("A and B must have the same dtype")
    P, H, W = A.shape
    R, h, w = B.shape
    dev = A.device
    out = torch.empty((2, P, R), device=dev, dtype=torch.int64)  # one allocation for both count matrices
    inter, union = out[0], out[1]
    nb = lib().sola_mask_iou_scratch_bytes(P, R, H, W)
    # the scratch is reused from call to call (same stream: the calls are ordered): the de-dup loop calls this once per SAM2
    # iteration with 10-35 MB of masks, where an allocation costs as much as the kernel
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    scratch = _IOU_SCRATCH.get(key)
    if scratch is None or scratch.numel() < nb:
        scratch = _IOU_SCRATCH[key] = torch.empty(nb, device=dev, dtype=torch.uint8)
    check(lib().sola_mask_iou_matrix(ptr(A), ptr(B), _elem_type(A), P, R, H, W, h, w, ptr(inter), ptr(union),
                                     ptr(scratch), scratch.numel(), current_stream(dev)), "sola_mask_iou_matrix")
    return inter, union


def iou_from_counts(inter, union):
    """python-float division; an empty union counts as IoU 1.0 (seg_utils.py:139-142)."""
    if union == 0:
        return 1.0
    return inter / union


@torch.no_grad()
def compute_mask_iou(maskA, maskB):
    """seg_utils.py:128-142: maskA, maskB (H,W) {0,1} -> float."""
    inter, union = mask_iou_matrix(maskA.unsqueeze(0), maskB.unsqueeze(0))
    i, u = torch.stack([inter[0, 0], union[0, 0]]).tolist()
    return iou_from_counts(i, u)


@torch.no_grad()
def compute_masklet_iou(maskletA, maskletB, device=None):
    """seg_utils.py:109-125: one ratio over all frames of two (T,H,W) masklets."""
    if device is not None:
        maskletA, maskletB = maskletA.to(device), maskletB.to(device)
    T, H, W = maskletA.shape
    inter, union = mask_iou_matrix(maskletA.reshape(1, T * H, W), maskletB.reshape(1, T * H, W))
    i, u = torch.stack([inter[0, 0], union[0, 0]]).tolist()
    return iou_from_counts(i, u)


@torch.no_grad()
def dedup_batch(masklets, prompt_ids, prompts, miou_thresh, reshape=False):
    """Greedy filtering of untracked prompts by the newly tracked masklets (generate_tokens_grid.py:252-278).

    masklets: dict prompt_id -> (T,H,W) {0,1} CUDA tensor at the comparison resolution (after reshape_masklet), or,
    with ``reshape=True``, at the tracker's native resolution — the bilinear resample + threshold of
    generate_tokens_grid.py:248-250 is then fused into the pack launch and no resized masklet is materialised;
    prompt_ids: new tracks in batch order; prompts: list of dicts (``status``, ``frame_idx``, ``segmentation``
    (h,w) array/tensor) mutated in place exactly like the reference.  All P x R intersections come from two pack
    launches and one pair launch and a single host copy; the order-dependent greedy decision runs on the host.
    """
    todo = [r for r, info in enumerate(prompts) if info["status"] == 0]
    if not todo or not prompt_ids:
        return 0
    first = masklets[prompt_ids[0]]
    dev = first.device
    T, H, W = first.shape
    A = torch.stack([masklets[pid] for pid in prompt_ids]).reshape(len(prompt_ids) * T, H, W)
    segs = [torch.as_tensor(prompts[r]["segmentation"]) for r in todo]
    B = torch.stack(segs).to(dev)
    if B.dtype not in (torch.uint8, torch.bool, torch.float32):
        B = (B != 0).to(torch.uint8)
    if reshape:
        a_bits, a_area, (H, W) = pack_masklet_bilinear(A)
    else:
        a_bits, a_area = pack_masks(A)
    b_bits, b_area = pack_masks(B, (H, W))
    frames = torch.tensor([prompts[r]["frame_idx"] for r in todo], dtype=torch.int32, device=dev)
    inter, union = pair_counts(a_bits, a_area, b_bits, b_area, T=T, a_frame=frames)
    inter, union = inter.cpu().tolist(), union.cpu().tolist()
    n_filtered = 0
    for p, pid in enumerate(prompt_ids):
        for c, r in enumerate(todo):
            info = prompts[r]
            if info["status"] > 0:
                continue
            iou = iou_from_counts(inter[p][c], union[p][c])
            if iou > miou_thresh:
                info["status"] = 2
                info["filtered_by"] = pid
                info["filtered_iou"] = iou
                n_filtered += 1
    return n_filtered


# ----------------------------------------------------------------------------------------------------------------
# masklet rows next to the predicate: reshape_masklet, per-frame metrics, part-ness, RLE merge
# ----------------------------------------------------------------------------------------------------------------
def default_target_shape(h, w):
    """seg_utils.py:154-156."""
    return (540, 960) if h < w else (960, 540)


def pack_masklet_bilinear(masklet, target_shape=None, logits=False):
    """[N,h,w] {0,1} -> (bits int32 [N,words], area int64 [N], (H,W)): bilinear resample + `> 0.5` + bit-pack in one
    pass over the source (seg_utils.py:145-160 without the fp32 [N,H,W] intermediate).  ``logits=True``: the input is
    the tracker's float32 mask logits and `(logits > 0).float()` (generate_tokens_grid.py:215-222) is applied on read."""
    masklet = _prep(masklet)
    if logits and masklet.dtype != torch.float32:
        raise SolaError("logits must be float32")
    n, h, w = masklet.shape
    H, W = default_target_shape(h, w) if target_shape is None else target_shape
    words = lib().sola_mask_words(H, W)
    bits = torch.empty((n, words), device=masklet.device, dtype=torch.int32)
    area = torch.empty((n,), device=masklet.device, dtype=torch.int64)
    check(lib().sola_mask_bilinear_pack(ptr(masklet), 2 if logits else _elem_type(masklet), n, h, w, H, W, ptr(bits), ptr(area),
                                        current_stream(masklet.device)), "sola_mask_bilinear_pack")
    return bits, area, (H, W)


def unpack_masks(bits, H, W, dtype=torch.float32):
    """bits [n,words] -> {0,1} images [n,H,W] of ``dtype`` (float32 or uint8)."""
    require_cuda(bits)
    n = bits.shape[0]
    out = torch.empty((n, H, W), device=bits.device, dtype=dtype)
    check(lib().sola_mask_unpack(ptr(bits), n, H, W, ptr(out), _elem_type(out), current_stream(bits.device)),
          "sola_mask_unpack")
    return out


def reshape_masklet(masklet, target_shape=None, logits=False):
    """seg_utils.py:145-160: (N,h,w) {0,1} -> (N,H',W') float32 {0,1}."""
    bits, _, (H, W) = pack_masklet_bilinear(masklet, target_shape, logits)
    return unpack_masks(bits, H, W, torch.float32)


@torch.no_grad()
def frame_counts(pred_masks, gt_masks):
    """(T,H,W) x (T,H,W) -> int64 [T,3] on the host: (intersection, n_pred, n_gt) per frame, from two pack launches,
    one pair launch and one copy (the reference does five .item() syncs per frame, utils.py:146-151)."""
    pred_masks, gt_masks = _prep(pred_masks), _prep(gt_masks)
    if pred_masks.shape != gt_masks.shape:
        raise SolaError(f"masklets differ in shape: {tuple(pred_masks.shape)} vs {tuple(gt_masks.shape)}")
    T = pred_masks.shape[0]
    a_bits, a_area = pack_masks(pred_masks)
    b_bits, b_area = pack_masks(gt_masks)
    frames = torch.arange(T, dtype=torch.int32, device=pred_masks.device)
    inter, _ = pair_counts(a_bits, a_area, b_bits, b_area, T=T, a_frame=frames)
    return torch.stack([inter[0], a_area, b_area], 1).cpu()


@torch.no_grad()
def masklet_counts_matrix(pred_masklets, gt_masklets):
    """(P,T,H,W) x (G,T,H,W) -> int64 [P,G,T,3] (intersection, n_pred, n_gt) for every (pred track, GT object, frame):
    the whole `for prompt_id ... for gt_anno_id ... for t` nest of generate_tokens_grid.py:252-264 in one pair launch."""
    pred_masklets, gt_masklets = _prep(pred_masklets), _prep(gt_masklets)
    P, T, H, W = pred_masklets.shape
    G = gt_masklets.shape[0]
    if tuple(gt_masklets.shape[1:]) != (T, H, W):
        raise SolaError("pred and GT masklets differ in (T,H,W)")
    a_bits, a_area = pack_masks(pred_masklets.reshape(P * T, H, W))
    b_bits, b_area = pack_masks(gt_masklets.reshape(G * T, H, W))
    frames = torch.arange(T, dtype=torch.int32, device=pred_masklets.device).repeat(G)
    inter, _ = pair_counts(a_bits, a_area, b_bits, b_area, T=T, a_frame=frames)  # [P, G*T]
    out = torch.stack([inter.view(P, G, T), a_area.view(P, 1, T).expand(P, G, T), b_area.view(1, G, T).expand(P, G, T)], -1)
    return out.cpu()


def metrics_from_counts(counts):
    """utils.py:146-168 on the integer counts [T,3] -> (precision, recall, iou) float32 [T] CPU tensors."""
    T = len(counts)
    precision, recall, iou = torch.zeros(T).float(), torch.zeros(T).float(), torch.zeros(T).float()
    for t, (intersection, n_pred, n_gt) in enumerate(counts.tolist() if hasattr(counts, "tolist") else counts):
        union = n_pred + n_gt - intersection
        iou[t] = 1.0 if union == 0 else intersection / union
        if n_pred == 0 and n_gt == 0:
            precision[t], recall[t] = 1.0, 1.0
        elif n_pred == 0 and n_gt > 0:
            precision[t], recall[t] = 1.0, 0.0
        elif n_pred > 0 and n_gt == 0:
            precision[t], recall[t] = 0.0, 1.0
        else:
            precision[t], recall[t] = intersection / n_pred, intersection / n_gt
    return precision, recall, iou


@torch.no_grad()
def compute_mask_metrics(pred_masks, gt_masks, reduction="mean"):
    """utils.py:131-174: (T,H,W) x (T,H,W) -> precision, recall, iou (0-d float32 tensors, or [T] with 'none')."""
    if reduction not in ("mean", "none"):
        raise ValueError(f"Invalid reduction method: {reduction}")
    precision, recall, iou = metrics_from_counts(frame_counts(pred_masks, gt_masks))
    if reduction == "mean":
        return precision.mean(), recall.mean(), iou.mean()
    return precision, recall, iou


def J_from_counts(counts):
    """evaluator.py:227-237."""
    import numpy as np
    Js = []
    for intersection, n_pred, n_gt in counts.tolist():
        union = n_pred + n_gt - intersection
        Js.append(1.0 if union == 0 else intersection / union)
    return np.mean(Js)


def F_from_counts(counts):
    """evaluator.py:239-247: tp / fp / fn over all frames."""
    tp, n_pred, n_gt = counts.sum(0).tolist()
    fp, fn = n_pred - tp, n_gt - tp
    if tp == 0:
        return 0.0
    precision = tp / (tp + fp)
    recall = tp / (tp + fn)
    return 2 * precision * recall / (precision + recall)


def compute_J(pred_masklet, gt_masklet):
    return J_from_counts(frame_counts(pred_masklet, gt_masklet))


def compute_F(pred_masklet, gt_masklet):
    return F_from_counts(frame_counts(pred_masklet, gt_masklet))


def compute_JF(pred_masklet, gt_masklet):
    """J, F and (J+F)/2 from ONE counting pass (evaluator.py:196-199 runs two)."""
    c = frame_counts(pred_masklet, gt_masklet)
    J, F = float(J_from_counts(c)), float(F_from_counts(c))
    return J, F, (J + F) / 2


@torch.no_grad()
def compute_P(part_masks, full_mask):
    """utils.py:177-192: part-ness |part & full| / |part| as float32 [N] on the masks' device (0/0 -> nan as there)."""
    part_masks, full_mask = _prep(part_masks), _prep(full_mask)
    if _elem_type(part_masks) != _elem_type(full_mask):
        full_mask = full_mask.to(part_masks.dtype)
    a_bits, a_area = pack_masks(part_masks)
    b_bits, b_area = pack_masks(full_mask.unsqueeze(0))
    inter, _ = pair_counts(a_bits, a_area, b_bits, b_area)
    return inter[:, 0].to(torch.float32) / a_area.to(torch.float32)


def _rle_cum(rle, limit):
    """Inclusive prefix sums (uint32) of one RLE dict's run lengths; compressed strings are parsed by the library's
    host helper (sola_rle_string_to_cum), uncompressed lists by numpy."""
    import ctypes
    import numpy as np
    counts = rle["counts"]
    if isinstance(counts, str):
        counts = counts.encode("ascii")
    if isinstance(counts, (bytes, bytearray)):
        buf = np.empty(max(1, len(counts)), np.uint32)
        n = lib().sola_rle_string_to_cum(bytes(counts), len(counts), ctypes.c_void_p(buf.ctypes.data), len(buf), limit)
        if n < 0:
            check(int(n), "sola_rle_string_to_cum")
        return buf[:n]
    c = np.cumsum(np.asarray(counts, dtype=np.int64))
    if len(c) and (c[-1] > limit or np.any(np.diff(c) < 0) or c[0] < 0):
        raise SolaError("rle_merge_or: runs are negative or exceed the image")
    return c.astype(np.uint32)


@torch.no_grad()
def rle_merge_or(rle_lists, device, packed=False):
    """OR of K RLE masklets (each a list of T per-frame COCO RLE dicts, non-dict = missing frame) decoded on the GPU:
    dataloader.py:326-369 (rle_masklet_decode + np.logical_or).  Only the run-length strings are parsed on the host.
    Returns uint8 [T,h,w] (or (bits, area, (h,w)) with ``packed``)."""
    import numpy as np
    K = len(rle_lists)
    if K == 0:
        raise SolaError("rle_merge_or: no masklets")
    T = len(rle_lists[0])
    size = next((tuple(r["size"]) for rl in rle_lists for r in rl if isinstance(r, dict)), None)
    if size is None:
        raise SolaError("rle_merge_or: every frame is missing")
    h, w = size
    cums, off = [], [0]
    for f in range(T):
        for k in range(K):
            r = rle_lists[k][f] if f < len(rle_lists[k]) else None
            if isinstance(r, dict):
                if tuple(r["size"]) != (h, w):
                    raise SolaError(f"rle_merge_or: frame size {tuple(r['size'])} != {(h, w)}")
                c = _rle_cum(r, h * w)
                cums.append(c)
                off.append(off[-1] + len(c))
            else:
                off.append(off[-1])
    cum = np.concatenate(cums) if cums else np.zeros(1, np.uint32)
    if len(cum) == 0:
        cum = np.zeros(1, np.uint32)
    dev = torch.device(device)
    cum_t = torch.from_numpy(cum.view(np.int32)).to(dev)
    off_t = torch.tensor(off, dtype=torch.int64, device=dev)
    stream = current_stream(dev)
    if packed:
        words = lib().sola_mask_words(h, w)
        bits = torch.empty((T, words), device=dev, dtype=torch.int32)
        area = torch.empty((T,), device=dev, dtype=torch.int64)
        check(lib().sola_rle_fill_or(ptr(cum_t), ptr(off_t), T, K, h, w, None, ptr(bits), ptr(area), stream), "sola_rle_fill_or")
        return bits, area, (h, w)
    out = torch.empty((T, h, w), device=dev, dtype=torch.uint8)
    check(lib().sola_rle_fill_or(ptr(cum_t), ptr(off_t), T, K, h, w, ptr(out), None, None, stream), "sola_rle_fill_or")
    return out
