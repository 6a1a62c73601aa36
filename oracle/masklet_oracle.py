"""CPU ORACLE for the masklet rows next to the IoU predicate.  TEST INFRASTRUCTURE ONLY.

numpy / python restatement of
  * ``track_generation/seg_utils.py:145-160``  ``reshape_masklet``  (``F.interpolate(mode='bilinear')`` then ``> 0.5``),
  * ``track_generation/utils.py:131-174``      ``compute_mask_metrics`` (per-frame precision / recall / IoU),
  * ``track_generation/utils.py:177-192``      ``compute_P`` (part-ness = |part & full| / |part|, float32 division),
  * ``evaluator.py:227-247``                   ``compute_J`` / ``compute_F``,
  * ``dataloader.py:305-369``                  RLE masklet decode + OR-merge of the selected tracks.
Only ``tests/`` (and ``__graft_entry__.smoke()`` / ``bench.py``'s cpu_baseline leg) may import this file.

Parity status:
  * reshape_masklet / compute_mask_metrics-style counts / compute_P / compute_J / compute_F: PINNED by
    ``tests/golden/masklet_golden.npz`` (made by ``tests/golden/gen_masklet_golden.py`` from the imported reference
    functions; ``evaluator.compute_J/F`` and ``utils.compute_*`` are copied-by-behaviour there because their modules
    import cv2 / pycocotools which this image lacks — the generator stubs those imports, see the script).
  * RLE decode: **parity unpinned** — the algorithm lives in pycocotools (``maskApi.c`` rleFrString / rleDecode,
    requirements.txt pins ``pycocotools==2.0.8``), which is absent here; it is restated from the published format and
    checked by encode -> decode round trips and hand-written vectors only.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


# ---------------------------------------------------------------------------------------------- bilinear resample
def _fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64; the double rounding of the sum
    (53 then 24 bits) can differ from a true fma only on exact 2^-29-ties, which the tests' inputs never produce."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def bilinear_index_weights(in_size, out_size):
    """ATen ``compute_source_index_and_lambda`` (UpSampleKernel.cpp) for align_corners=False:
    ``src = max(scale*(dst+0.5)-0.5, 0)`` with ``scale = float32(in)/out`` and the multiply-add fused (what the AVX2 /
    AVX512 CPU build executes — verified bit-for-bit against ``F.interpolate`` on random float images in the test);
    ``i0 = min(int(src), in-1)``, ``l1 = clamp(src - i0, 0, 1)``, ``i1 = i0 + (i0 < in-1)``, ``l0 = 1 - l1``."""
    if in_size == out_size:
        idx = np.arange(out_size, dtype=np.int64)
        return idx, idx, np.ones(out_size, f32), np.zeros(out_size, f32)
    scale = f32(in_size) / f32(out_size)
    dst = np.arange(out_size).astype(f32) + f32(0.5)
    src = _fma32(np.full(out_size, scale, f32), dst, np.full(out_size, f32(-0.5)))
    src = np.where(src < 0, f32(0), src).astype(f32)
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    l1 = np.clip((src - i0.astype(f32)).astype(f32), f32(0), f32(1)).astype(f32)
    i1 = i0 + (i0 < in_size - 1)
    l0 = (f32(1) - l1).astype(f32)
    return i0, i1, l0, l1


def _wsum(w0, a, w1, b, rounding):
    """w0*a + w1*b in float32 under the three roundings a compiler may pick for it."""
    w0, w1 = np.broadcast_to(w0, a.shape), np.broadcast_to(w1, b.shape)
    if rounding == "fma":      # fma(w0, a, round(w1*b)) — what the ATen CPU build here executes
        return _fma32(w0, a, (w1 * b).astype(f32))
    if rounding == "fma_rev":  # fma(w1, b, round(w0*a))
        return _fma32(w1, b, (w0 * a).astype(f32))
    if rounding == "plain":    # both products rounded, then the sum
        return ((w0 * a).astype(f32) + (w1 * b).astype(f32)).astype(f32)
    raise ValueError(rounding)


def bilinear_resize(x, H, W, rounding="fma"):
    """x [n,h,w] float32 -> [n,H,W] float32, ``F.interpolate(x[None], size=(H,W), mode='bilinear')[0]``:
    ``out = l0y*(l0x*v00 + l1x*v01) + l1y*(l0x*v10 + l1x*v11)``.  ``rounding`` selects how the two-term sums are
    rounded (see _wsum); for {0,1} images the ``> 0.5`` decision is the same under all of them (tested)."""
    x = np.asarray(x, f32)
    y0, y1, ly0, ly1 = bilinear_index_weights(x.shape[1], H)
    x0, x1, lx0, lx1 = bilinear_index_weights(x.shape[2], W)

    def row(r):
        xr = x[:, r]
        return _wsum(lx0, xr[:, :, x0], lx1, xr[:, :, x1], rounding)

    t, u = row(y0), row(y1)
    return _wsum(ly0[None, :, None], t, ly1[None, :, None], u, rounding)


def default_target_shape(h, w):
    """seg_utils.py:154-156: landscape -> (540, 960), otherwise (960, 540)."""
    return (540, 960) if h < w else (960, 540)


def reshape_masklet(masklet, target_shape=None):
    """seg_utils.py:145-160: [N,h,w] {0,1} -> [N,H',W'] float32 {0,1}."""
    masklet = np.asarray(masklet, f32)
    H, W = default_target_shape(*masklet.shape[1:]) if target_shape is None else target_shape
    return (bilinear_resize(masklet, H, W) > f32(0.5)).astype(f32)


# ---------------------------------------------------------------------------------------------- per-frame metrics
def frame_counts(pred, gt):
    """[T,H,W] x [T,H,W] -> int64 [T,3] (intersection, n_pred, n_gt): the exact integers behind every float32 sum of
    utils.py:147-151 / evaluator.py:230-232 (counts < 2^24)."""
    p = np.asarray(pred) != 0
    g = np.asarray(gt) != 0
    T = p.shape[0]
    return np.stack([(p & g).reshape(T, -1).sum(1), p.reshape(T, -1).sum(1), g.reshape(T, -1).sum(1)], 1).astype(np.int64)


def metrics_from_counts(counts):
    """utils.py:146-168 on integer counts: three float32 [T] vectors (python-float ratios stored into float32)."""
    T = len(counts)
    prec, rec, iou = np.zeros(T, f32), np.zeros(T, f32), np.zeros(T, f32)
    for t, (inter, n_pred, n_gt) in enumerate(np.asarray(counts).tolist()):
        union = n_pred + n_gt - inter
        iou[t] = 1.0 if union == 0 else inter / union
        if n_pred == 0 and n_gt == 0:
            prec[t], rec[t] = 1.0, 1.0
        elif n_pred == 0 and n_gt > 0:
            prec[t], rec[t] = 1.0, 0.0
        elif n_pred > 0 and n_gt == 0:
            prec[t], rec[t] = 0.0, 1.0
        else:
            prec[t], rec[t] = inter / n_pred, inter / n_gt
    return prec, rec, iou


def compute_mask_metrics(pred, gt, reduction="mean"):
    """utils.py:131-174.  ``mean`` reduces with torch's float32 mean, as the reference does."""
    prec, rec, iou = metrics_from_counts(frame_counts(pred, gt))
    if reduction == "none":
        return prec, rec, iou
    if reduction != "mean":
        raise ValueError(f"Invalid reduction method: {reduction}")
    import torch
    return tuple(float(torch.from_numpy(v).mean()) for v in (prec, rec, iou))


def J_from_counts(counts):
    """evaluator.py:227-237: mean over frames (float64, np.mean) of inter/union, empty union -> 1."""
    js = []
    for inter, n_pred, n_gt in np.asarray(counts).tolist():
        union = n_pred + n_gt - inter
        js.append(1.0 if union == 0 else inter / union)
    return float(np.mean(js))


def F_from_counts(counts):
    """evaluator.py:239-247: one precision / recall over ALL frames' pixels."""
    c = np.asarray(counts).sum(0).tolist()
    tp, fp, fn = c[0], c[1] - c[0], c[2] - c[0]
    if tp == 0:
        return 0.0
    precision, recall = tp / (tp + fp), tp / (tp + fn)
    return 2 * precision * recall / (precision + recall)


def compute_J(pred, gt):
    return J_from_counts(frame_counts(pred, gt))


def compute_F(pred, gt):
    return F_from_counts(frame_counts(pred, gt))


def compute_P(part_masks, full_mask):
    """utils.py:177-192: float32 ``(part @ full) / part.sum`` -> [N] float32 (0/0 = nan, as in the reference)."""
    p = np.asarray(part_masks) != 0
    g = np.asarray(full_mask) != 0
    n = p.shape[0]
    inter = (p & g[None]).reshape(n, -1).sum(1).astype(f32)
    area = p.reshape(n, -1).sum(1).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (inter / area).astype(f32)


# ---------------------------------------------------------------------------------------------- COCO RLE
def rle_string_to_counts(s):
    """pycocotools ``rleFrString``: 5 data bits + continuation bit per char (offset 48), sign extension from bit 4 of
    the last char, and every run from the 4th on is stored as a delta to the run two places back."""
    if isinstance(s, (bytes, bytearray)):
        s = s.decode("ascii")
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_counts_to_string(counts):
    """pycocotools ``rleToString`` (inverse of the above)."""
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def mask_to_counts(mask):
    """uint8 [h,w] -> run lengths over the column-major flattening, starting with a (possibly empty) run of zeros."""
    flat = (np.asarray(mask) != 0).astype(np.uint8).T.reshape(-1)
    change = np.flatnonzero(np.diff(flat)) + 1
    runs = np.diff(np.concatenate([[0], change, [flat.size]])).tolist()
    if flat.size and flat[0] == 1:
        runs = [0] + runs
    return runs


def rle_decode(rle):
    """pycocotools ``rleDecode``: {size:[h,w], counts: str|bytes|list} -> uint8 [h,w]."""
    h, w = rle["size"]
    counts = rle["counts"]
    if not isinstance(counts, (list, tuple, np.ndarray)):
        counts = rle_string_to_counts(counts)
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T.copy()


def masklet_decode(rle_list):
    """dataloader.py:353-369: per-frame RLE dicts (non-dict = missing frame -> zeros of the last decoded size)."""
    frames, h, w = [], 0, 0
    for r in rle_list:
        if isinstance(r, dict):
            m = rle_decode(r)
            h, w = m.shape
            frames.append(m)
        else:
            frames.append(None)
    return np.stack([f if f is not None else np.zeros((h, w), np.uint8) for f in frames], 0)


def merge_selected(rle_lists, preds):
    """dataloader.py:326-351 (get_sam2_masklet) without the file IO: OR of the decoded masklets with pred > 0, in
    listing order; all-zero [T,h,w] when nothing is selected; None when there are no tracks at all."""
    merged = None
    for rles, p in zip(rle_lists, preds):
        if p < 1 and merged is not None:
            continue
        if p > 0:
            m = masklet_decode(rles)
            merged = m if merged is None else np.logical_or(merged, m)
        elif merged is None:
            h, w = rles[0]["size"]
            merged = np.zeros((len(rles), h, w), np.uint8)
    return merged
