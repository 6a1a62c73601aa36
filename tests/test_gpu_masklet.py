"""HIP masklet rows (reshape_masklet, per-frame metrics, part-ness, RLE merge) against the reference-made
golden vectors and the CPU oracle, through the C-ABI.  Everything here is bit-exact."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import masklet_cases as mc  # noqa: E402
from oracle import iou_oracle  # noqa: E402
from oracle import masklet_oracle as mo  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def su():
    from sola_amd import seg_utils
    return seg_utils


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "masklet_golden.npz"))


def unpack(packed, w):
    return np.unpackbits(packed, axis=-1)[..., :w]


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8])
def test_reshape_masklet_small_cases(su, golden, dtype):
    for i, (n, h, w, H, W) in enumerate(mc.SMALL_SHAPES):
        x = unpack(golden[f"small{i}_in"], w)
        want = unpack(golden[f"small{i}_out"], W)
        y = su.reshape_masklet(dev(x, dtype), (H, W))
        assert y.dtype == torch.float32 and tuple(y.shape) == (n, H, W)
        np.testing.assert_array_equal(y.cpu().numpy().astype(np.uint8), want)
        bits, area, hw = su.pack_masklet_bilinear(dev(x, dtype), (H, W))
        assert hw == (H, W)
        np.testing.assert_array_equal(area.cpu().numpy(), want.reshape(n, -1).sum(1))
        np.testing.assert_array_equal(su.unpack_masks(bits, H, W, torch.uint8).cpu().numpy(), want)


def test_reshape_masklet_all_tap_patterns(su, golden):
    y = su.reshape_masklet(dev(mc.parity_images(37, 53), torch.float32), (54, 96))
    np.testing.assert_array_equal(y.cpu().numpy().astype(np.uint8), unpack(golden["parity_small_out"], 96))


@pytest.mark.parametrize("i", range(len(mc.PRODUCTION_SHAPES)))
def test_reshape_masklet_production_shapes(su, golden, i):
    """Full-size frames, default target rule, against the reference's output (areas + sha256 of the packed result)."""
    h, w = mc.PRODUCTION_SHAPES[i]
    x = mc.production_masklet(h, w, seed=i)
    y = su.reshape_masklet(dev(x, torch.float32))
    n, _, _, H, W = golden[f"prod{i}_shape"]
    assert tuple(y.shape) == (n, H, W)
    y = y.cpu().numpy()
    assert set(np.unique(y)) <= {0.0, 1.0}
    np.testing.assert_array_equal(y.reshape(n, -1).sum(1).astype(np.int64), golden[f"prod{i}_area"])
    assert mc.digest(y) == str(golden[f"prod{i}_digest"])
    y8 = su.reshape_masklet(dev(x))  # uint8 source
    assert mc.digest(y8.cpu().numpy()) == str(golden[f"prod{i}_digest"])


def test_reshape_masklet_random_shapes_vs_oracle(su):
    rng = np.random.default_rng(11)
    for _ in range(20):
        n, h, w, H, W = (int(v) for v in (rng.integers(1, 5), rng.integers(1, 150), rng.integers(1, 150),
                                          rng.integers(1, 150), rng.integers(1, 150)))
        x = (rng.random((n, h, w)) < 0.5).astype(np.uint8)
        want = mo.reshape_masklet(x, (H, W))
        got = su.reshape_masklet(dev(x, torch.float32), (H, W)).cpu().numpy()
        np.testing.assert_array_equal(got, want, err_msg=f"{(n, h, w, H, W)}")


def test_reshape_masklet_from_logits(su):
    """(logits > 0).float() of generate_tokens_grid.py:215-222 folded into the read."""
    rng = np.random.default_rng(12)
    for (n, h, w, H, W) in [(3, 64, 96, 54, 96), (2, 50, 31, 33, 47), (2, 480, 854, 540, 960)]:
        logits = rng.standard_normal((n, h, w)).astype(np.float32) * 3
        logits[0, :4] = 0.0  # exactly zero is background
        want = mo.reshape_masklet((logits > 0).astype(np.float32), (H, W))
        got = su.reshape_masklet(dev(logits), (H, W), logits=True).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    with pytest.raises(Exception):
        su.reshape_masklet(dev(np.zeros((1, 4, 4), np.uint8)), (4, 4), logits=True)


def test_mask_metrics_J_F_P_golden(su, golden):
    T, H, W = golden["met_shape"]
    pred, gt = unpack(golden["met_pred"], W), unpack(golden["met_gt"], W)
    p, g = dev(pred, torch.float32), dev(gt, torch.float32)
    np.testing.assert_array_equal(su.frame_counts(p, g).numpy(), mo.frame_counts(pred, gt))
    none = su.compute_mask_metrics(p, g, reduction="none")
    np.testing.assert_array_equal(torch.stack(none).numpy(), golden["met_none"])
    mean = su.compute_mask_metrics(p, g)
    assert all(v.dim() == 0 and v.dtype == torch.float32 for v in mean)
    np.testing.assert_array_equal(np.array([float(v) for v in mean], np.float32), golden["met_mean"])
    assert float(su.compute_J(p, g)) == float(golden["J"])
    assert float(su.compute_F(p, g)) == float(golden["F"])
    assert su.compute_F(p, 1 - p) == 0.0
    assert float(su.compute_J(p * 0, g * 0)) == 1.0
    J, F, JF = su.compute_JF(p, g)
    assert (J, F, JF) == (float(golden["J"]), float(golden["F"]), (float(golden["J"]) + float(golden["F"])) / 2)
    with pytest.raises(ValueError):
        su.compute_mask_metrics(p, g, reduction="sum")
    parts, full = unpack(golden["P_parts"], W), unpack(golden["P_full"], W)
    P = su.compute_P(dev(parts, torch.float32), dev(full, torch.float32))
    assert P.is_cuda and P.dtype == torch.float32
    P, want = P.cpu().numpy(), golden["P"]
    np.testing.assert_array_equal(np.isnan(P), np.isnan(want))
    np.testing.assert_array_equal(P[~np.isnan(P)], want[~np.isnan(want)])


def test_counts_matrix_vs_oracle_full_size(su):
    """P pred tracks x G GT objects x T frames at 540x960 in one pair launch (generate_tokens_grid.py:252-264)."""
    P, G, T, H, W = 3, 2, 5, 540, 960
    pred = np.stack([mc.blob_masklet(T, H, W, 20 + p) for p in range(P)])
    gt = np.stack([mc.blob_masklet(T, H, W, 40 + g) for g in range(G)])
    got = su.masklet_counts_matrix(dev(pred), dev(gt)).numpy()
    assert got.shape == (P, G, T, 3)
    for p in range(P):
        for g in range(G):
            np.testing.assert_array_equal(got[p, g], mo.frame_counts(pred[p], gt[g]))


def make_rles(tracks, compressed):
    out = []
    for t in tracks:
        frames = []
        for f in t:
            counts = mo.mask_to_counts(f)
            frames.append({"size": list(f.shape), "counts": mo.rle_counts_to_string(counts) if compressed else counts})
        out.append(frames)
    return out


@pytest.mark.parametrize("shape", [(3, 12, 9), (4, 37, 53), (3, 540, 960), (2, 960, 540)])
@pytest.mark.parametrize("compressed", [False, True])
def test_rle_merge_or_vs_oracle(su, shape, compressed):
    T, h, w = shape
    tracks = [mc.blob_masklet(T, h, w, s) for s in (1, 2, 3)]
    tracks[0][0, 0, 0] = 1  # a mask whose first run of zeros is empty
    rles = make_rles(tracks, compressed)
    rles[1][1] = None  # missing frame -> zeros
    want = mo.merge_selected(rles, [1, 1, 1]) != 0
    got = su.rle_merge_or(rles, "cuda")
    assert got.dtype == torch.uint8 and tuple(got.shape) == (T, h, w)
    np.testing.assert_array_equal(got.cpu().numpy() != 0, want)
    bits, area, hw = su.rle_merge_or(rles, "cuda", packed=True)
    assert hw == (h, w)
    np.testing.assert_array_equal(area.cpu().numpy(), want.reshape(T, -1).sum(1))
    np.testing.assert_array_equal(su.unpack_masks(bits, h, w, torch.uint8).cpu().numpy() != 0, want)
    one = su.rle_merge_or(rles[2:], "cuda")  # a single track is a plain decode
    np.testing.assert_array_equal(one.cpu().numpy(), mo.masklet_decode(rles[2]))


def test_dedup_with_fused_reshape_matches_two_step(su):
    """generate_tokens_grid.py:248-278 with the tracker's native-resolution masklets: fused resample+pack == resample,
    then pack; and both equal the oracle's greedy loop on the oracle-resampled masklets."""
    T, h, w = 4, 480, 854
    ids = [5, 9, 2]
    native = {pid: mc.blob_masklet(T, h, w, 60 + pid) for pid in ids}
    rng = np.random.default_rng(5)

    def prompts():
        out = []
        for r in range(24):
            pid = ids[r % 3]
            f = int(rng.integers(0, T))
            seg = mo.reshape_masklet(native[pid][f:f + 1], (270, 480))[0].astype(np.uint8)  # a prompt at another scale
            if r % 4 == 0:
                seg = np.roll(seg, int(rng.integers(1, 60)), axis=1)
            out.append({"status": 0 if r % 7 else 1, "frame_idx": f, "segmentation": seg})
        return out

    rng = np.random.default_rng(5)
    pa = prompts()
    rng = np.random.default_rng(5)
    pb = prompts()
    rng = np.random.default_rng(5)
    pc = prompts()
    na = su.dedup_batch({p: dev(native[p], torch.float32) for p in ids}, ids, pa, 0.7, reshape=True)
    resized = {p: su.reshape_masklet(dev(native[p], torch.float32)) for p in ids}
    nb = su.dedup_batch(resized, ids, pb, 0.7)
    want = {p: mo.reshape_masklet(native[p]) for p in ids}
    nc = iou_oracle.dedup_batch(want, ids, pc, 0.7)
    assert na == nb == nc and na > 0
    for a, b, c in zip(pa, pb, pc):
        assert a["status"] == b["status"] == c["status"]
        assert a.get("filtered_by") == b.get("filtered_by") == c.get("filtered_by")
        assert a.get("filtered_iou") == b.get("filtered_iou") == c.get("filtered_iou")


def test_dataset_merged_masklet_on_device_matches_host_decoder(su, tmp_path):
    """dataloader.py:305-351 through TrackDataset: the GPU decode+OR equals the host decoder for every selection,
    including "nothing selected" (zeros) and compressed-string RLEs."""
    import json

    from sola_amd import data as sdata
    data_root, track_root = tmp_path / "data", tmp_path / "tracks"
    os.makedirs(data_root / "mevis" / "valid_u")
    meta = {"videos": {"vidA": {"frames": ["00000", "00001", "00002"], "expressions": {"0": {"exp": "a cat", "anno_id": [3]}}}}}
    json.dump(meta, open(data_root / "mevis" / "valid_u" / "meta_expressions.json", "w"))
    mdir = track_root / "grid_tracks" / "mevis" / "valid_u" / "sam2_masklets" / "vidA"
    tdir = track_root / "grid_tracks" / "mevis" / "valid_u" / "sam2_object_tokens" / "vidA"
    os.makedirs(mdir), os.makedirs(tdir)
    for aid in (2, 5, 11):
        frames = mc.blob_masklet(3, 40, 64, aid)
        rle = [{"size": [40, 64], "counts": mo.rle_counts_to_string(mo.mask_to_counts(f))} for f in frames]
        json.dump({"anno_id": aid, "prompt_type": "X", "rle": rle}, open(mdir / f"{aid:05d}.json", "w"))
        np.save(tdir / f"{aid:05d}.npy", np.zeros((3, 256), np.float32))
    split = {"data_name": "mevis", "data_type": "valid_u", "sam2_output_dirs": "grid_tracks", "batch_size": 1}
    ds = sdata.TrackDataset(split, str(data_root), str(track_root))
    for preds in ([1, 0, 1], [0, 1, 0], [1, 1, 1], [0, 0, 0]):
        host = ds.merged_masklet("vidA", "0", np.array(preds))
        devm = ds.merged_masklet("vidA", "0", np.array(preds), device="cuda")
        assert devm.is_cuda and devm.dtype == torch.uint8
        np.testing.assert_array_equal(devm.cpu().numpy() != 0, np.asarray(host) != 0)


@pytest.mark.parametrize("staged", [1, 2])
def test_float_images_that_are_not_binary_keep_aten_arithmetic(su, staged):
    """elem_type float32 with values outside {0,1}: the `> 0.5` mask equals the oracle's float resample (ATen's rule),
    also when only ONE pixel of the image is soft - with the direct kernel (default for float sources) and with the
    LDS-staged one, whose blocks fall back to float taps when they meet a non-binary value."""
    from sola_amd import _lib
    _lib.lib().sola_tune(b"bilinear_staged", staged)
    try:
        _soft_mask_checks(su)
    finally:
        _lib.lib().sola_tune(b"bilinear_staged", 1)


def _soft_mask_checks(su):
    rng = np.random.default_rng(21)
    for (n, h, w, H, W) in [(2, 64, 96, 54, 96), (2, 120, 160, 90, 128), (1, 480, 856, 540, 960)]:
        x = rng.random((n, h, w)).astype(np.float32)           # soft masks
        want = (mo.bilinear_resize(x, H, W) > np.float32(0.5)).astype(np.float32)
        got = su.reshape_masklet(dev(x), (H, W)).cpu().numpy()
        assert (got != want).mean() < 2e-5  # values within 1 ulp of 0.5 may round differently under the fused sums
        y = (rng.random((n, h, w)) < 0.5).astype(np.float32)
        y[0, h // 2, w // 2] = 0.75                             # one soft pixel in a binary image
        want = (mo.bilinear_resize(y, H, W) > np.float32(0.5)).astype(np.float32)
        got = su.reshape_masklet(dev(y), (H, W)).cpu().numpy()
        assert (got != want).mean() < 2e-5
        # and the soft pixel matters: the all-binary version of the same image differs somewhere near it
        y2 = y.copy(); y2[0, h // 2, w // 2] = 1.0
        got2 = su.reshape_masklet(dev(y2), (H, W)).cpu().numpy()
        np.testing.assert_array_equal(got2, mo.reshape_masklet(y2, (H, W)))
