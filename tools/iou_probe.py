"""Mask IoU at the de-dup loop's call sizes (P = 4 tracks x R prompts, 540x960 uint8 resident in HBM): the one-launch kernel
(sola_tune iou_fused 1, default) against pack + pair (iou_fused 0): wall per call, in-library kernel time, fraction of the HBM peak.

    python tools/iou_probe.py [reps = 200]        (rocprofv3 --kernel-trace --stats -- python3 tools/iou_probe.py 50 for per-kernel times)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import iou_oracle  # noqa: E402  (checker only)
from sola_amd import _lib, seg_utils  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H, W, P = 540, 960, 4
rng = np.random.default_rng(0)


def rects(n):
    out = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        y0, x0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
        out[i, y0:y0 + rng.integers(8, H // 2), x0:x0 + rng.integers(8, W // 2)] = 1
    return out


for R in (16, 64, 256):
    A, Bm = rects(P), rects(R)
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
    ri, ru = iou_oracle.iou_matrix(A, Bm) if R <= 64 else (None, None)
    for mode in (1, 0):
        _lib.check(_lib.lib().sola_tune(b"iou_fused", mode), "tune")
        inter, union = seg_utils.mask_iou_matrix(a, b)
        if ri is not None:
            assert np.array_equal(inter.cpu().numpy(), ri) and np.array_equal(union.cpu().numpy(), ru)
        for _ in range(10):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        _lib.profile_enable(True)
        _lib.profile_read(True)
        for _ in range(reps):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        prof = _lib.profile_read(True)
        _lib.profile_enable(False)
        k_us = (prof["iou_pack"]["ms"] + prof["iou_pair"]["ms"]) / reps * 1e3
        nbytes = (P + R) * H * W
        print(json.dumps({"R": R, "path": "one launch" if mode else "pack + pair", "call_us_wall": round(wall * 1e6, 1), "kernels_us": round(k_us, 1),
                          "launches": (prof["iou_pack"]["launches"] + prof["iou_pair"]["launches"]) // reps,
                          "frac_hbm_kernels": round(nbytes / (k_us * 1e-6) / 8e12, 3), "frac_hbm_wall": round(nbytes / wall / 8e12, 3)}), flush=True)
_lib.check(_lib.lib().sola_tune(b"iou_fused", 1), "tune")
