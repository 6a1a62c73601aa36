#!/usr/bin/env python3
"""A/B of the two mask-IoU paths at the de-dup loop's call sizes (P=4 x R prompts, 540x960 uint8): the one-launch fused kernel
(sola_tune iou_fused 2 = forced) against pack + pair (iou_fused 0): wall time per call and in-library kernel time."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from sola_amd import _lib, seg_utils

rng = np.random.default_rng(0)
H, W, P = 540, 960, 4
for R in (16, 32, 64, 128, 256):
    a = torch.from_numpy((rng.uniform(size=(P, H, W)) < 0.3).astype(np.uint8)).cuda()
    b = torch.from_numpy((rng.uniform(size=(R, H, W)) < 0.3).astype(np.uint8)).cuda()
    row = {"R": R}
    ref = None
    for tag, mode in (("pack_pair", 0), ("fused", 2)):
        _lib.check(_lib.lib().sola_tune(b"iou_fused", mode), "tune")
        inter, union = seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        if ref is None:
            ref = (inter.clone(), union.clone())
        assert torch.equal(inter, ref[0]) and torch.equal(union, ref[1])
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        _lib.profile_enable(True); _lib.profile_read(True)
        for _ in range(50):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        prof = _lib.profile_read(True); _lib.profile_enable(False)
        k_us = (prof["iou_pack"]["ms"] + prof["iou_pair"]["ms"]) / 50 * 1e3
        row[tag] = {"call_us_wall": round(wall * 1e6, 1), "kernels_us": round(k_us, 1)}
    _lib.check(_lib.lib().sola_tune(b"iou_fused", 1), "tune")
    print(json.dumps(row), flush=True)
