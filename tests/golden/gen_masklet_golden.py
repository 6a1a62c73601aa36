"""Generate tests/golden/masklet_golden.npz by running the REAL reference functions of the §8f rows.

Run in the authoring container only (needs /root/reference):

    python tests/golden/gen_masklet_golden.py

Imported unmodified: ``track_generation/seg_utils.reshape_masklet`` (:145-160), ``track_generation/utils.
compute_mask_metrics`` (:131-174) and ``compute_P`` (:177-192), ``evaluator.Evaluator.compute_J/compute_F`` (:227-247,
called unbound — they do not touch ``self``).  Those modules import pycocotools / cv2 / imageio at module top, which
this image lacks and which the functions above never use; empty module objects are registered under those names for
the import only.  Inputs come from tests/masklet_cases.py (seeded), so the large cases store only areas + digests.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "track_generation"))

import masklet_cases as mc  # noqa: E402

for name in ("pycocotools", "pycocotools.mask", "cv2", "imageio"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]

import seg_utils as ref_seg  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference track_generation/utils.py)
from evaluator import Evaluator  # noqa: E402  (reference)

torch.set_num_threads(8)


def main():
    store = {}
    # (1) reshape_masklet, small shapes stored in full (inputs and outputs bit-packed along the last axis)
    for i, (n, h, w, H, W) in enumerate(mc.SMALL_SHAPES):
        x = mc.blob_masklet(n, h, w, seed=100 + i)
        y = ref_seg.reshape_masklet(torch.from_numpy(x).float(), target_shape=(H, W)).numpy()
        assert y.shape == (n, H, W) and set(np.unique(y)) <= {0.0, 1.0}
        store[f"small{i}_in"] = np.packbits(x, axis=-1)
        store[f"small{i}_out"] = np.packbits(y.astype(np.uint8), axis=-1)
        store[f"small{i}_shape"] = np.array([n, h, w, H, W])
    # parity images at small size through the default-free path as well (all 16 tap patterns)
    x = mc.parity_images(37, 53)
    y = ref_seg.reshape_masklet(torch.from_numpy(x).float(), target_shape=(54, 96)).numpy()
    store["parity_small_out"] = np.packbits(y.astype(np.uint8), axis=-1)
    # (2) production shapes with the default target rule: per-frame areas + digest of the packed output
    for i, (h, w) in enumerate(mc.PRODUCTION_SHAPES):
        x = mc.production_masklet(h, w, seed=i)
        y = ref_seg.reshape_masklet(torch.from_numpy(x).float()).numpy()
        store[f"prod{i}_shape"] = np.array([x.shape[0], h, w, y.shape[1], y.shape[2]])
        store[f"prod{i}_area"] = y.reshape(y.shape[0], -1).sum(1).astype(np.int64)
        store[f"prod{i}_digest"] = np.array(mc.digest(y))
        print("reshape", (h, w), "->", y.shape[1:], store[f"prod{i}_digest"])
    # (3) compute_mask_metrics / compute_J / compute_F on masklet pairs with empty-frame edge cases
    T, H, W = 8, 54, 96
    pred = mc.blob_masklet(T, H, W, seed=7)
    gt = mc.blob_masklet(T, H, W, seed=8)
    gt[0] = pred[0]            # identical frame
    pred[1] = 0; gt[1] = 0     # both empty
    pred[2] = 0                # pred empty, gt not
    gt[3] = 0                  # gt empty, pred not
    store["met_pred"], store["met_gt"] = np.packbits(pred, axis=-1), np.packbits(gt, axis=-1)
    pt, gtt = torch.from_numpy(pred).float(), torch.from_numpy(gt).float()
    store["met_mean"] = np.array([float(v) for v in ref_utils.compute_mask_metrics(pt, gtt)], np.float32)
    store["met_none"] = np.stack([v.numpy() for v in ref_utils.compute_mask_metrics(pt, gtt, reduction="none")])
    store["J"] = np.array(float(Evaluator.compute_J(None, pt, gtt)))
    store["F"] = np.array(float(Evaluator.compute_F(None, pt, gtt)))
    store["F_disjoint"] = np.array(float(Evaluator.compute_F(None, pt, 1 - pt)))  # tp == 0 -> 0.0
    store["J_empty"] = np.array(float(Evaluator.compute_J(None, pt * 0, gtt * 0)))
    # (4) compute_P: parts vs a full mask, including an empty part (0/0 -> nan)
    parts = mc.blob_masklet(7, H, W, seed=9)   # includes an empty and a full frame
    full = mc.blob_masklet(1, H, W, seed=10)[0]
    store["P_parts"], store["P_full"] = np.packbits(parts, axis=-1), np.packbits(full, axis=-1)
    store["P"] = ref_utils.compute_P(torch.from_numpy(parts).float(), torch.from_numpy(full).float()).numpy()
    store["met_shape"] = np.array([T, H, W])
    np.savez_compressed(os.path.join(HERE, "masklet_golden.npz"), **store)
    print("wrote masklet_golden.npz", os.path.getsize(os.path.join(HERE, "masklet_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
