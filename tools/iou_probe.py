#!/usr/bin/env python3
"""Mask-IoU call at the de-dup loop's sizes, fused one-launch path vs pack + pair: wall time per call and (under
`rocprofv3 --kernel-trace --stats -- python3 tools/iou_probe.py`) the kernels' own durations."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, seg_utils

H, W, P = 540, 960, 4
rng = np.random.default_rng(0)
for R in (16, 64, 256):
    a = torch.from_numpy((rng.uniform(size=(P, H, W)) < 0.3).astype(np.uint8)).cuda()
    b = torch.from_numpy((rng.uniform(size=(R, H, W)) < 0.3).astype(np.uint8)).cuda()
    row = {"R": R}
    for fused in (1, 0):
        _lib.check(_lib.lib().sola_tune(b"iou_fused", fused), "tune")
        for _ in range(5):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        row["fused_us" if fused else "pack_pair_us"] = round((time.perf_counter() - t0) / 100 * 1e6, 1)
    print(json.dumps(row))
