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


def mask_iou_matrix(A, B):
    """A [P,H,W], B [R,h,w] (resampled to H x W) -> (inter, union) int64 [P,R] in one library call."""
    A, B = _prep(A), _prep(B)
    if _elem_type(A) != _elem_type(B):
        raise SolaError("A and B must have the same dtype")
    P, H, W = A.shape
    R, h, w = B.shape
    dev = A.device
    inter = torch.empty((P, R), device=dev, dtype=torch.int64)
    union = torch.empty((P, R), device=dev, dtype=torch.int64)
    nb = lib().sola_mask_iou_scratch_bytes(P, R, H, W)
    scratch = torch.empty(nb, device=dev, dtype=torch.uint8)
    check(lib().sola_mask_iou_matrix(ptr(A), ptr(B), _elem_type(A), P, R, H, W, h, w, ptr(inter), ptr(union),
                                     ptr(scratch), nb, current_stream(dev)), "sola_mask_iou_matrix")
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
def dedup_batch(masklets, prompt_ids, prompts, miou_thresh):
    """Greedy filtering of untracked prompts by the newly tracked masklets (generate_tokens_grid.py:252-278).

    masklets: dict prompt_id -> (T,H,W) {0,1} CUDA tensor at the comparison resolution (after reshape_masklet);
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
