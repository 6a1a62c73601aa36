"""oracle/masklet_oracle.py against the golden vectors made from the imported reference (gen_masklet_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import masklet_cases as mc  # noqa: E402
from oracle import masklet_oracle as mo  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "masklet_golden.npz"))


def unpack(packed, w):
    return np.unpackbits(packed, axis=-1)[..., :w]


def test_reshape_small_cases_bit_exact(golden):
    for i, (n, h, w, H, W) in enumerate(mc.SMALL_SHAPES):
        assert tuple(golden[f"small{i}_shape"]) == (n, h, w, H, W)
        x = unpack(golden[f"small{i}_in"], w)
        np.testing.assert_array_equal(x, mc.blob_masklet(n, h, w, seed=100 + i))
        want = unpack(golden[f"small{i}_out"], W)
        got = mo.reshape_masklet(x, (H, W))
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got.astype(np.uint8), want)


def test_reshape_all_tap_patterns_small(golden):
    got = mo.reshape_masklet(mc.parity_images(37, 53), (54, 96))
    np.testing.assert_array_equal(got.astype(np.uint8), unpack(golden["parity_small_out"], 96))


@pytest.mark.parametrize("i", range(len(mc.PRODUCTION_SHAPES)))
def test_reshape_production_shapes_digest(golden, i):
    h, w = mc.PRODUCTION_SHAPES[i]
    x = mc.production_masklet(h, w, seed=i)
    y = mo.reshape_masklet(x)  # default target rule
    n, _, _, H, W = golden[f"prod{i}_shape"]
    assert y.shape == (n, H, W) and (H, W) == mo.default_target_shape(h, w)
    np.testing.assert_array_equal(y.reshape(n, -1).sum(1).astype(np.int64), golden[f"prod{i}_area"])
    assert mc.digest(y) == str(golden[f"prod{i}_digest"])


@pytest.mark.parametrize("shape", [(720, 1280, 540, 960), (480, 854, 540, 960), (360, 640, 540, 960), (1280, 720, 960, 540),
                                   (37, 53, 54, 96), (17, 9, 40, 33)])
def test_threshold_decision_does_not_depend_on_sum_rounding(shape):
    """All 16 tap patterns at every output pixel: fma / reversed fma / unfused sums give the same `> 0.5` mask, so the
    HIP kernel's choice (and ATen's, which differs between its CPU code paths) cannot change a bit."""
    h, w, H, W = shape
    x = mc.parity_images(h, w).astype(np.float32)
    ref = mo.bilinear_resize(x, H, W, "fma") > np.float32(0.5)
    for r in ("fma_rev", "plain"):
        np.testing.assert_array_equal(mo.bilinear_resize(x, H, W, r) > np.float32(0.5), ref)


def test_binary_decisions_match_aten_on_random_shapes():
    """The index/weight rule against this machine's ATen CPU kernel, on white-noise masks of random sizes (the digest
    tests above pin the same thing against the reference function itself)."""
    rng = np.random.default_rng(3)
    for _ in range(25):
        n, h, w, H, W = (int(v) for v in (rng.integers(1, 4), rng.integers(1, 120), rng.integers(1, 120),
                                          rng.integers(1, 120), rng.integers(1, 120)))
        x = (rng.random((n, h, w)) < 0.5).astype(np.float32)
        want = (F.interpolate(torch.from_numpy(x)[None], size=(H, W), mode="bilinear") > 0.5)[0].numpy()
        np.testing.assert_array_equal(mo.reshape_masklet(x, (H, W)) > 0, want)


def test_mask_metrics_J_F_P(golden):
    T, H, W = golden["met_shape"]
    pred, gt = unpack(golden["met_pred"], W), unpack(golden["met_gt"], W)
    p, r, i = mo.compute_mask_metrics(pred, gt, "none")
    np.testing.assert_array_equal(np.stack([p, r, i]), golden["met_none"])
    np.testing.assert_array_equal(np.array(mo.compute_mask_metrics(pred, gt), np.float32), golden["met_mean"])
    assert mo.compute_J(pred, gt) == float(golden["J"])
    assert mo.compute_F(pred, gt) == float(golden["F"])
    assert mo.compute_F(pred, 1 - pred) == float(golden["F_disjoint"]) == 0.0
    assert mo.compute_J(pred * 0, gt * 0) == float(golden["J_empty"]) == 1.0
    parts, full = unpack(golden["P_parts"], W), unpack(golden["P_full"], W)
    got = mo.compute_P(parts, full)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(golden["P"]))
    np.testing.assert_array_equal(got[~np.isnan(got)], golden["P"][~np.isnan(golden["P"])])
    assert np.isnan(golden["P"]).sum() == 1  # the empty part


def test_rle_round_trip_and_known_vectors():
    # hand-written vectors of the published format: 3x2 mask, column-major runs
    m = np.array([[0, 1], [1, 1], [0, 0]], np.uint8)  # columns: (0,1,0), (1,1,0)
    assert mo.mask_to_counts(m) == [1, 1, 1, 2, 1]
    np.testing.assert_array_equal(mo.rle_decode({"size": [3, 2], "counts": [1, 1, 1, 2, 1]}), m)
    assert mo.mask_to_counts(np.ones((2, 2), np.uint8)) == [0, 4]
    # small counts map to single chars '0'+c; 32 needs a continuation char; negative deltas use the sign bit
    assert mo.rle_counts_to_string([1, 1, 1]) == "111"
    assert mo.rle_string_to_counts("111") == [1, 1, 1]
    for counts in ([0, 4], [5, 40, 3, 2, 100000, 1, 7], [1000, 3, 2, 1, 900, 2, 1], [0, 1, 0, 1]):
        assert mo.rle_string_to_counts(mo.rle_counts_to_string(counts)) == counts
    rng = np.random.default_rng(0)
    for _ in range(10):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        m = mc.blob_masklet(1, h, w, int(rng.integers(1 << 30)))[0]
        rle = {"size": [h, w], "counts": mo.rle_counts_to_string(mo.mask_to_counts(m))}
        np.testing.assert_array_equal(mo.rle_decode(rle), m)


def test_merge_selected_rules():
    h, w, T = 12, 9, 3
    tracks = [mc.blob_masklet(T, h, w, s) for s in (1, 2, 3)]
    rles = [[{"size": [h, w], "counts": mo.mask_to_counts(f)} for f in t] for t in tracks]
    rles[1][1] = None  # a missing frame decodes to zeros
    tracks[1][1] = 0
    np.testing.assert_array_equal(mo.merge_selected(rles, [1, 0, 1]) != 0, (tracks[0] | tracks[2]) != 0)
    np.testing.assert_array_equal(mo.merge_selected(rles, [0, 1, 1]) != 0, (tracks[1] | tracks[2]) != 0)
    z = mo.merge_selected(rles, [0, 0, 0])
    assert z.shape == (T, h, w) and not z.any()
    assert mo.merge_selected([], []) is None
