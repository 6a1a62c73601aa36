#!/usr/bin/env python3
"""Mask-IoU de-dup workload (BASELINE config C3): P=4 new-track masks vs R prompts at 540x960, uint8 and
float32 masks, with and without the nearest resample of the prompts.  Prints one JSON object per case:
HIP-event time of the pack kernel (the HBM-bound one), its GB/s against the 8 TB/s peak, the pair kernel time, the
end-to-end call rate, and the CPU oracle's rate on the same inputs.  Counts are checked against the oracle."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sola_amd import _lib, seg_utils
from oracle import iou_oracle

def rects(rng, n, H, W):
    out = np.zeros((n, H, W), np.uint8)
    for i in range(n):
        y0, x0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
        out[i, y0:y0 + rng.integers(8, H // 2), x0:x0 + rng.integers(8, W // 2)] = 1
    return out

rng = np.random.default_rng(0)
H, W, P = 540, 960, 4
for R, (h, w), dt in [(16, (540, 960), np.uint8), (64, (540, 960), np.uint8), (256, (540, 960), np.uint8), (256, (540, 960), np.float32),
                      (256, (720, 1280), np.uint8)]:
    A = rects(rng, P, H, W).astype(dt); Bm = rects(rng, R, h, w).astype(dt)
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
    inter, union = seg_utils.mask_iou_matrix(a, b); torch.cuda.synchronize()
    if R <= 64:
        ri, ru = iou_oracle.iou_matrix(A, iou_oracle.nearest_resize(Bm, H, W))
        assert np.array_equal(inter.cpu().numpy(), ri) and np.array_equal(union.cpu().numpy(), ru)
    _lib.profile_enable(True); _lib.profile_read(True)
    t0 = time.perf_counter(); reps = 50
    for _ in range(reps): seg_utils.mask_iou_matrix(a, b)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
    prof = _lib.profile_read(True); _lib.profile_enable(False)
    pk, pr = prof["iou_pack"], prof["iou_pair"]
    src_bytes = (A.nbytes + Bm.nbytes)
    t_pack = pk["ms"] / reps * 1e-3
    tc = time.perf_counter(); iou_oracle.iou_matrix(A[:1], iou_oracle.nearest_resize(Bm[:8], H, W)); cpu_pair = (time.perf_counter() - tc) / 8
    print(json.dumps({"workload": f"mask IoU P={P} R={R} {h}x{w}->{H}x{W} {np.dtype(dt).name}", "pairs": P * R,
                      "pack_us": round(t_pack * 1e6, 1), "pack_GBps": round(src_bytes / t_pack / 1e9, 1),
                      "pack_frac_of_8TBps": round(src_bytes / t_pack / 8e12, 3), "pair_us": round(pr["ms"] / reps * 1e3, 1),
                      "call_us_wall": round(wall * 1e6, 1), "pairs_per_s": round(P * R / wall), "cpu_oracle_pairs_per_s": round(1 / cpu_pair, 1)}))
