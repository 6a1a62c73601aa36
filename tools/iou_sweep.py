"""Sweep of the one-launch mask-IoU kernel block shapes (sola_tune iou_shape = 16 * G + nb): kernel us per call, P = 4, 540x960 uint8; counts checked
against pack + pair.  python tools/iou_sweep.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import _lib, seg_utils
H, W, P = 540, 960, 4
rng = np.random.default_rng(0)
def rects(n):
    out = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        y0, x0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
        out[i, y0:y0 + rng.integers(8, H // 2), x0:x0 + rng.integers(8, W // 2)] = 1
    return out
for R in (16, 64, 256):
    a, b = torch.from_numpy(rects(P)).cuda(), torch.from_numpy(rects(R)).cuda()
    _lib.check(_lib.lib().sola_tune(b"iou_fused", 0), "t")
    ref = seg_utils.mask_iou_matrix(a, b)
    _lib.check(_lib.lib().sola_tune(b"iou_fused", 1), "t")
    for G, nb in ((1,1),(1,2),(1,4),(2,1),(2,2),(2,4),(4,1),(4,2),(4,3)):
        _lib.check(_lib.lib().sola_tune(b"iou_shape", 16*G+nb), "t")
        out = seg_utils.mask_iou_matrix(a, b)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), (R, G, nb)
        for _ in range(20): seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        _lib.profile_enable(True); _lib.profile_read(True)
        for _ in range(200): seg_utils.mask_iou_matrix(a, b)
        torch.cuda.synchronize()
        prof = _lib.profile_read(True); _lib.profile_enable(False)
        print(R, G, nb, round(prof["iou_pack"]["ms"] / 200 * 1e3, 2), "us", flush=True)
