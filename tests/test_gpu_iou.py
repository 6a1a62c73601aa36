"""Mask-IoU parity (integer work: bit-exact) through the C ABI: golden vectors of the reference's
compute_mask_iou / F.interpolate(nearest) / de-dup loop, plus oracle comparisons on seeded random masks and
size-independent properties at the full 540x960 resolution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import iou_oracle  # noqa: E402
from sola_amd import seg_utils  # noqa: E402


def unpack(a, w):
    return np.unpackbits(a, axis=-1)[..., :w]


def cuda(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_golden_pairs(iou_golden, dtype):
    A = unpack(iou_golden["pair_A"], 960).astype(dtype)
    B = unpack(iou_golden["pair_B"], 960).astype(dtype)
    inter, union = seg_utils.mask_iou_matrix(cuda(A), cuda(B))
    inter, union = inter.cpu().numpy(), union.cpu().numpy()
    got = np.array([[seg_utils.iou_from_counts(int(inter[p, r]), int(union[p, r])) for r in range(B.shape[0])]
                    for p in range(A.shape[0])])
    np.testing.assert_array_equal(got, iou_golden["pair_iou"])  # identical doubles
    np.testing.assert_array_equal(got > 0.7, iou_golden["pair_iou"] > 0.7)
    assert got[3, 5] == 1.0  # empty vs empty
    # single-pair drop-in (seg_utils.py:128-142)
    assert seg_utils.compute_mask_iou(cuda(A[0]), cuda(B[1])) == iou_golden["pair_iou"][0, 1]


def test_tie_is_not_above_threshold(iou_golden):
    a = np.zeros((10, 10), np.float32)
    b = np.zeros((10, 10), np.float32)
    a[0, :10] = 1
    b[0, :7] = 1
    v = seg_utils.compute_mask_iou(cuda(a), cuda(b))
    assert v == float(iou_golden["tie_iou"]) and not (v > 0.7)


def test_masklet_iou(iou_golden):
    v = seg_utils.compute_masklet_iou(cuda(iou_golden["masklet_A"].astype(np.float32)),
                                      cuda(iou_golden["masklet_B"].astype(np.float32)), "cuda")
    assert v == float(iou_golden["masklet_iou"])


def test_nearest_resize_matches_aten(iou_golden):
    """Pack with resampling == pack of the oracle-resampled mask, for every non-integer scale in the fixture."""
    rng = np.random.default_rng(3)
    for i, (h, w, Ho, Wo) in enumerate(iou_golden["resize_sizes"]):
        h, w, Ho, Wo = int(h), int(w), int(Ho), int(Wo)
        src = (rng.uniform(size=(3, h, w)) < 0.4).astype(np.uint8)
        # the index maps are pinned by the fixture (F.interpolate), the oracle reproduces them (CPU test)
        ref = src[:, iou_golden[f"resize{i}_row"][:, None], iou_golden[f"resize{i}_col"][None, :]]
        bits, area = seg_utils.pack_masks(cuda(src), (Ho, Wo))
        rbits, rarea = seg_utils.pack_masks(cuda(np.ascontiguousarray(ref)))
        assert torch.equal(bits, rbits), (h, w, Ho, Wo)
        np.testing.assert_array_equal(area.cpu().numpy(), ref.reshape(3, -1).sum(axis=1))
        assert torch.equal(area, rarea)


def test_dedup_loop_golden(iou_golden):
    tracks = unpack(iou_golden["dedup_tracks"], 960)
    segs = unpack(iou_golden["dedup_segs"], 640)
    ids = [int(v) for v in iou_golden["dedup_ids"]]
    masklets = {pid: cuda(tracks[i].astype(np.float32)) for i, pid in enumerate(ids)}
    prompts = [{"status": int(s), "frame_idx": int(f), "segmentation": segs[r]}
               for r, (s, f) in enumerate(zip(iou_golden["dedup_status_in"], iou_golden["dedup_frame_idx"]))]
    n = seg_utils.dedup_batch(masklets, ids, prompts, 0.7)
    assert n == int(iou_golden["dedup_n_filtered"])
    np.testing.assert_array_equal([p["status"] for p in prompts], iou_golden["dedup_status_out"])
    np.testing.assert_array_equal([p.get("filtered_by", -1) for p in prompts], iou_golden["dedup_filtered_by"])
    np.testing.assert_array_equal([p.get("filtered_iou", -1.0) for p in prompts], iou_golden["dedup_filtered_iou"])


@pytest.mark.parametrize("P,R,H,W,h,w", [(4, 16, 540, 960, 540, 960), (3, 7, 37, 53, 37, 53), (5, 9, 64, 100, 48, 77),
                                         (1, 1, 8, 8, 8, 8), (9, 33, 30, 31, 60, 62)])
def test_random_masks_vs_oracle(P, R, H, W, h, w):
    rng = np.random.default_rng(P * R + H)
    A = (rng.uniform(size=(P, H, W)) < 0.3).astype(np.uint8)
    B = (rng.uniform(size=(R, h, w)) < 0.5).astype(np.uint8)
    inter, union = seg_utils.mask_iou_matrix(cuda(A), cuda(B))
    Br = iou_oracle.nearest_resize(B, H, W)
    ri, ru = iou_oracle.iou_matrix(A, Br)
    np.testing.assert_array_equal(inter.cpu().numpy(), ri)
    np.testing.assert_array_equal(union.cpu().numpy(), ru)


def test_full_size_properties():
    """R=256 prompts at 540x960: symmetry, self-IoU, inclusion-exclusion, and a checksum against popcounts."""
    H, W, R = 540, 960, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    B = (torch.rand((R, H, W), device="cuda", generator=g) < 0.35).to(torch.uint8)
    A = B[:4].clone()
    A[1, :100] = 0
    inter, union = seg_utils.mask_iou_matrix(A, B)
    inter_t, union_t = seg_utils.mask_iou_matrix(B[:8], A)
    assert torch.equal(inter[:, :8], inter_t.t()) and torch.equal(union[:, :8], union_t.t())
    area_b = B.view(R, -1).sum(dim=1, dtype=torch.int64)
    area_a = A.view(4, -1).sum(dim=1, dtype=torch.int64)
    assert torch.equal(inter + union, area_a[:, None] + area_b[None, :])  # |A|+|B| = |A&B| + |A|B|
    assert inter[0, 0] == union[0, 0] == area_b[0]  # IoU(A,A) = 1
    assert torch.all(inter <= torch.minimum(area_a[:, None], area_b[None, :]))
    assert inter[1, 1] == A[1].sum()  # A[1] is a subset of B[1]


@pytest.mark.parametrize("P,R,H,W", [(4, 16, 540, 960), (1, 1, 540, 960), (3, 64, 540, 960), (4, 7, 64, 96), (2, 5, 960, 540), (4, 300, 128, 256),
                                     (4, 33, 1080, 1920), (2, 3, 8, 4), (4, 129, 36, 100)])
def test_fused_one_launch_path_equals_pack_and_pair(P, R, H, W):
    """uint8 masks at the comparison resolution with P <= 4 (the de-dup loop's calls) take the one-launch kernel (round 6: no memset,
    per-block count rows + a ticket per prompt group, iou.hip); its counts are the oracle's and the three-kernel path's, including
    empty masks, a mask equal to its partner, a last chunk that is mostly out of range (1080 x 1920: 127 chunks; 8 x 4: one word)
    and prompt counts that leave the last group ragged."""
    from sola_amd import _lib

    rng = np.random.default_rng(P * 1000 + R)
    A = (rng.uniform(size=(P, H, W)) < 0.3).astype(np.uint8)
    B = (rng.uniform(size=(R, H, W)) < 0.5).astype(np.uint8) * rng.integers(1, 255, size=(R, H, W)).astype(np.uint8)  # any non-zero byte counts
    B[0] = 0
    if R > 2:
        B[2] = A[0]
    A[P - 1, : H // 2] = 0
    ri, ru = iou_oracle.iou_matrix(A, (B != 0).astype(np.uint8))
    outs = []
    for fused in (2, 0):  # 2 = the fused kernel for any R (by default it serves calls with up to 32 prompts)
        _lib.check(_lib.lib().sola_tune(b"iou_fused", fused), "sola_tune")
        inter, union = seg_utils.mask_iou_matrix(cuda(A), cuda(B))
        outs.append((inter.cpu().numpy(), union.cpu().numpy()))
    _lib.check(_lib.lib().sola_tune(b"iou_fused", 1), "sola_tune")
    for inter, union in outs:
        np.testing.assert_array_equal(inter, ri)
        np.testing.assert_array_equal(union, ru)
    # repeated calls reuse the scratch and the library's ticket ring: every ticket must be back at zero
    for _ in range(3):
        inter, union = seg_utils.mask_iou_matrix(cuda(A), cuda(B))
        np.testing.assert_array_equal(inter.cpu().numpy(), ri)


@pytest.mark.parametrize("R", [1, 16, 70, 300])
def test_packed_pair_words_at_the_largest_counts(R):
    """Round 6: masks of fewer than 2^19 pixels accumulate [arrivals | |A| | |B| | inter] per (track, prompt) pair in ONE 64-bit word (19-bit
    fields, an atomic add that returns the old word; sola_tune "iou_packed" 0 = the ticket form).  540 x 960 = 518 400 pixels is 1.1 % below
    the field's range: all-ones masks put every field at its largest value; and the two forms agree on random masks, call after call (the
    words go back to zero)."""
    from sola_amd import _lib

    H, W = 540, 960
    rng = np.random.default_rng(R)
    A = np.ones((4, H, W), np.uint8)
    B = np.ones((R, H, W), np.uint8)
    if R > 2:
        B[1] = (rng.uniform(size=(H, W)) < 0.5)
        A[2] = (rng.uniform(size=(H, W)) < 0.9)
    ri, ru = iou_oracle.iou_matrix(A, B)
    assert ri.max() == H * W
    for packed in (1, 0, 1):
        _lib.check(_lib.lib().sola_tune(b"iou_packed", packed), "sola_tune")
        try:
            for _ in range(3):
                inter, union = seg_utils.mask_iou_matrix(cuda(A), cuda(B))
                np.testing.assert_array_equal(inter.cpu().numpy(), ri)
                np.testing.assert_array_equal(union.cpu().numpy(), ru)
        finally:
            _lib.check(_lib.lib().sola_tune(b"iou_packed", 1), "sola_tune")


def test_one_launch_kernel_on_two_streams_at_once():
    """The one-launch kernel's tickets live in the library (a ring range per launch): calls in flight on two streams at the same
    time take disjoint ranges - 200 interleaved calls of different sizes, every result equal to the pack + pair path's."""
    from sola_amd import _lib

    rng = np.random.default_rng(7)
    sets = []
    for R in (16, 70):
        A = (rng.uniform(size=(4, 540, 960)) < 0.3).astype(np.uint8)
        B = (rng.uniform(size=(R, 540, 960)) < 0.4).astype(np.uint8)
        a, b = cuda(A), cuda(B)
        _lib.check(_lib.lib().sola_tune(b"iou_fused", 0), "sola_tune")
        ref = seg_utils.mask_iou_matrix(a, b)
        _lib.check(_lib.lib().sola_tune(b"iou_fused", 1), "sola_tune")
        sets.append((a, b, ref))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for _ in range(100):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i].append(seg_utils.mask_iou_matrix(sets[i][0], sets[i][1]))
    torch.cuda.synchronize()
    for i in range(2):
        for inter, union in outs[i]:
            assert torch.equal(inter, sets[i][2][0]) and torch.equal(union, sets[i][2][1])


@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
@pytest.mark.parametrize("h,w,H,W", [(720, 1280, 540, 960), (480, 854, 540, 960), (1080, 1920, 540, 960), (100, 37, 64, 96), (270, 480, 540, 960),
                                     (33, 4100, 40, 64)])
def test_lds_staged_resample_pack_equals_per_pixel_pack(dtype, h, w, H, W):
    """The LDS-staged nearest-resample pack (rows of any alignment, uint8 and float32 sources, sources wider than its LDS rows
    fall back) writes the bits and areas of the per-pixel kernel, which the ATen index maps of the fixture pin."""
    from sola_amd import _lib

    rng = np.random.default_rng(h + w)
    src = ((rng.uniform(size=(3, h, w)) < 0.4) * rng.integers(1, 200, size=(3, h, w))).astype(dtype)
    outs = []
    for lds in (2, 0):  # 2 = the LDS-staged kernel for every source width (by default unaligned rows keep the gather kernel)
        _lib.check(_lib.lib().sola_tune(b"pack_resample_lds", lds), "sola_tune")
        bits, area = seg_utils.pack_masks(cuda(src), (H, W))
        outs.append((bits.clone(), area.clone()))
    _lib.check(_lib.lib().sola_tune(b"pack_resample_lds", 1), "sola_tune")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = iou_oracle.nearest_resize((src != 0).astype(np.uint8), H, W)
    np.testing.assert_array_equal(outs[0][1].cpu().numpy(), ref.reshape(3, -1).sum(axis=1))
