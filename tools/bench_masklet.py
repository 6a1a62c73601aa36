#!/usr/bin/env python3
"""Masklet rows next to the IoU predicate: reshape_masklet (bilinear -> >0.5 -> bit-pack), unpack, RLE
OR-merge.  One JSON object per case: HIP-event kernel time, GB/s of algorithmic bytes (source read once + packed bits /
image written once) against the 8 TB/s HBM peak, wall time per call, and the CPU oracle's time on a bounded sample."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sola_amd import _lib, seg_utils
from oracle import masklet_oracle as mo
import masklet_cases as mc


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_read(True)
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
    prof = _lib.profile_read(True); _lib.profile_enable(False)
    return prof["iou_pack"]["ms"] / reps * 1e-3, prof["iou_pack"]["launches"] // reps, wall


T = 64
for (h, w), dt in [((720, 1280), torch.float32), ((720, 1280), torch.uint8), ((480, 854), torch.float32), ((1080, 1920), torch.float32),
                   ((540, 960), torch.float32)]:
    x_np = mc.blob_masklet(T, h, w, 0)
    x = torch.from_numpy(x_np).cuda().to(dt)
    H, W = seg_utils.default_target_shape(h, w)
    bits, area, _ = seg_utils.pack_masklet_bilinear(x)
    t0 = time.perf_counter(); want = mo.reshape_masklet(x_np[:2]); cpu = (time.perf_counter() - t0) / 2
    assert np.array_equal(seg_utils.unpack_masks(bits[:2], H, W, torch.uint8).cpu().numpy(), want.astype(np.uint8))
    k, _, wall = timed(lambda: seg_utils.pack_masklet_bilinear(x))
    src = x.numel() * x.element_size(); out = bits.numel() * 4
    print(json.dumps({"workload": f"reshape_masklet fused pack T={T} {h}x{w}->{H}x{W} {str(dt)[6:]}", "kernel_us": round(k * 1e6, 1),
                      "GBps": round((src + out) / k / 1e9, 1), "frac_of_8TBps": round((src + out) / k / 8e12, 3),
                      "call_us_wall": round(wall * 1e6, 1), "frames_per_s": round(T / wall), "cpu_oracle_ms_per_frame": round(cpu * 1e3, 1)}))
    k, n, wall = timed(lambda: seg_utils.reshape_masklet(x))
    outb = T * H * W * 4
    print(json.dumps({"workload": f"reshape_masklet drop-in (float32 [T,{H},{W}] result) {str(dt)[6:]}", "kernels_us": round(k * 1e6, 1),
                      "launches": n, "GBps": round((src + 2 * out + outb) / k / 1e9, 1), "call_us_wall": round(wall * 1e6, 1)}))

# RLE OR-merge: K selected tracks of T frames at 540x960 -> uint8 [T,540,960]
T, K, h, w = 32, 4, 540, 960
tracks = [mc.blob_masklet(T, h, w, 10 + k) for k in range(K)]
rles = [[{"size": [h, w], "counts": mo.mask_to_counts(f)} for f in t] for t in tracks]
t0 = time.perf_counter(); want = mo.merge_selected(rles, [1] * K); cpu = time.perf_counter() - t0
got = seg_utils.rle_merge_or(rles, "cuda")
assert np.array_equal(got.cpu().numpy() != 0, want != 0)
t0 = time.perf_counter(); seg_utils.rle_merge_or(rles, "cuda"); torch.cuda.synchronize(); host_and_gpu = time.perf_counter() - t0
k, _, wall = timed(lambda: seg_utils.rle_merge_or(rles, "cuda"), reps=5)
print(json.dumps({"workload": f"RLE decode + OR-merge K={K} T={T} {h}x{w}", "kernel_us": round(k * 1e6, 1),
                  "out_GBps": round(T * h * w / k / 1e9, 1), "call_ms_wall_incl_host_parse": round(wall * 1e3, 2),
                  "cpu_oracle_ms": round(cpu * 1e3, 1)}))
