"""CPU ORACLE for the pairwise mask-IoU de-duplication predicate.  TEST INFRASTRUCTURE ONLY.

numpy restatement of ``track_generation/seg_utils.py:128-142`` (``compute_mask_iou``), ``:109-125``
(``compute_masklet_iou``), the nearest-neighbour prompt-mask resize in
``track_generation/generate_tokens_grid.py:269-272`` (``F.interpolate(mode='nearest')``) and the greedy
filtering loop ``generate_tokens_grid.py:252,266-278``.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg import this file; the product never does.

Parity status: PINNED by ``tests/golden/iou_golden.npz`` (made by ``tests/golden/gen_golden.py`` from the
imported reference functions and ``torch.nn.functional.interpolate``).

The reference sums float32 {0,1} tensors; the counts are < 2^24 so those sums are exact integers and the
IoU is an integer ratio evaluated in double precision (python float division).
"""
from __future__ import annotations

import numpy as np


def nearest_index(out_size, in_size):
    """Source index for every destination index, as ATen's ``nearest`` mode computes it:
    ``min(floor(dst * float32(in/out)), in - 1)`` with the scale and the product in float32."""
    if out_size == in_size:
        return np.arange(out_size, dtype=np.int64)
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = np.floor(dst * scale).astype(np.int64)
    return np.minimum(src, in_size - 1)


def nearest_resize(mask, H, W):
    """mask [..., h, w] -> [..., H, W] (generate_tokens_grid.py:272)."""
    h, w = mask.shape[-2:]
    iy = nearest_index(H, h)
    ix = nearest_index(W, w)
    return mask[..., iy[:, None], ix[None, :]]


def mask_counts(a, b):
    """(intersection, union) as exact integers: sum(A*B), sum(A+B) - sum(A*B) (seg_utils.py:137-138)."""
    a = np.asarray(a) != 0
    b = np.asarray(b) != 0
    inter = int(np.count_nonzero(a & b))
    union = int(np.count_nonzero(a)) + int(np.count_nonzero(b)) - inter
    return inter, union


def iou_from_counts(inter, union):
    """python-float division; empty union counts as identical masks (seg_utils.py:139-142)."""
    if union == 0:
        return 1.0
    return inter / union


def compute_mask_iou(a, b):
    return iou_from_counts(*mask_counts(a, b))


def compute_masklet_iou(a, b):
    """Whole-masklet [T,H,W] IoU (seg_utils.py:109-125): one ratio over all frames."""
    return iou_from_counts(*mask_counts(a, b))


def iou_matrix(A, B):
    """A [P,H,W], B [R,H,W] -> inter [P,R], union [P,R] int64."""
    P, R = A.shape[0], B.shape[0]
    inter = np.zeros((P, R), dtype=np.int64)
    union = np.zeros((P, R), dtype=np.int64)
    for p in range(P):
        for r in range(R):
            inter[p, r], union[p, r] = mask_counts(A[p], B[r])
    return inter, union


def dedup_batch(masklets, prompt_ids, prompts, miou_thresh):
    """Greedy filtering of untracked prompts by the newly tracked masklets (generate_tokens_grid.py:252-278).

    masklets: dict prompt_id -> [T,H,W] {0,1} array (already at the 540x960 comparison resolution);
    prompt_ids: the new tracks in batch order; prompts: list of dicts with ``status``, ``frame_idx``,
    ``segmentation`` [h,w]; mutated in place exactly like the reference.  Returns the number filtered.
    """
    n_filtered = 0
    for pid in prompt_ids:
        for info in prompts:
            if info["status"] > 0:
                continue
            pred = masklets[pid][info["frame_idx"]]
            H, W = pred.shape
            pm = nearest_resize(np.asarray(info["segmentation"]), H, W)
            iou = compute_mask_iou(pred, pm)
            if iou > miou_thresh:
                info["status"] = 2
                info["filtered_by"] = pid
                info["filtered_iou"] = iou
                n_filtered += 1
    return n_filtered
