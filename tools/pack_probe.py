#!/usr/bin/env python3
"""Nearest-resample mask pack 720x1280 -> 540x960 (256 masks): LDS-staged kernel vs per-pixel gather, HIP-event time."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, seg_utils
rng = np.random.default_rng(0)
for dt in (np.uint8, np.float32):
    for (h, w) in ((720, 1280), (1080, 1920), (480, 854)):
        n = 256 if dt == np.uint8 else 64
        src = torch.from_numpy((rng.uniform(size=(n, h, w)) < 0.4).astype(dt)).cuda()
        row = {"src": f"{n}x{h}x{w} {np.dtype(dt).name}", "bytes": int(src.numel() * src.element_size())}
        for lds in (1, 0):
            _lib.check(_lib.lib().sola_tune(b"pack_resample_lds", lds), "tune")
            for _ in range(3): seg_utils.pack_masks(src, (540, 960))
            torch.cuda.synchronize()
            _lib.profile_enable(True); _lib.profile_read(True)
            for _ in range(20): seg_utils.pack_masks(src, (540, 960))
            torch.cuda.synchronize()
            p = _lib.profile_read(True)["iou_pack"]; _lib.profile_enable(False)
            us = p["ms"] / 20 * 1e3
            row["lds_us" if lds else "gather_us"] = round(us, 1)
            row["lds_frac_hbm" if lds else "gather_frac_hbm"] = round(row["bytes"] / (us * 1e-6) / 8e12, 3)
        print(json.dumps(row))
