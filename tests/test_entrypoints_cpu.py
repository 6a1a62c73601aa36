"""Host glue of the entry points (no GPU): CLI/YAML contract, output-directory layout, the on-disk track format reader
and the COCO-RLE decoder."""
import json
import os

import numpy as np
import torch
import yaml

from sola_amd import config as sconfig
from sola_amd import data as sdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_overrides_and_layout(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    os.makedirs("configs/mevis")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "mevis", "default.yaml")))
    cfg["results"] = {"output_dir": "O/TRAIN", "eval_output_dir": "O/EVAL", "test_output_dir": "O/INF"}
    yaml.safe_dump(cfg, open("configs/mevis/default.yaml", "w"))
    c = sconfig.load_configs("train", ["--config", "mevis/default", "--lr_override", "0.5", "--flag", "--n", "3", "--s", "abc", "--b", "False"])
    assert c["lr_override"] == 0.5 and c["flag"] is True and c["n"] == 3 and c["s"] == "abc" and c["b"] is False
    assert c["results"]["output_dir"] == os.path.join("O/TRAIN", "default", "mevis") and os.path.isdir(c["results"]["output_dir"])
    c = sconfig.load_configs("inference", ["--config", "mevis/default", "--eval_weight_epoch", "7", "--eval_pred_threshold", "0.5"])
    assert c["eval"]["weight_path"] == os.path.join("O/TRAIN", "default", "mevis", "epoch_7.pth")
    assert c["results"]["test_output_dir"] == os.path.join("O/INF", "default", "mevis", "pred_threshold_05", "epoch_7")
    c = sconfig.load_configs("eval", ["--config", "mevis/default", "--eval_weight_epoch", "2"])
    assert c["results"]["eval_output_dir"].endswith(os.path.join("pred_threshold_05", "epoch_2"))
    assert c["model"]["lang_token_dim"] == 1024 and c["train"]["positive_weight"] == 1.5


def test_rle_roundtrip_and_known_string():
    rng = np.random.default_rng(0)
    for shape in [(7, 5), (1, 9), (16, 16)]:
        m = (rng.uniform(size=shape) < 0.4).astype(np.uint8)
        np.testing.assert_array_equal(sdata.rle_decode(sdata.rle_encode_uncompressed(m)), m)
    np.testing.assert_array_equal(sdata.rle_decode(sdata.rle_encode_uncompressed(np.ones((3, 4), np.uint8))), np.ones((3, 4)))
    # compressed string (chars = value + 48; runs after the 2nd are stored as a difference to the run two back):
    # [2,3,4,3] -> "2", "3", "4", 3-3=0 -> "2340": 2 zeros, 3 ones, 4 zeros, 3 ones, column-major on a 3x4 grid
    m = sdata.rle_decode({"size": [3, 4], "counts": "2340"})
    np.testing.assert_array_equal(m.T.reshape(-1), [0, 0, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1])
    assert sdata.rle_counts_from_string("2340") == [2, 3, 4, 3] and sdata.rle_counts_from_string("2343") == [2, 3, 4, 6]
    # multi-char / delta-coded run: 100 = 0b1100100 -> chars (4|0x20)+48='T', 3+48='3'
    assert sdata.rle_counts_from_string("T3") == [100]


def test_track_dataset_reads_the_on_disk_contract(tmp_path):
    data_root, track_root = tmp_path / "data", tmp_path / "tracks"
    os.makedirs(data_root / "mevis" / "valid_u")
    meta = {"videos": {"vidA": {"frames": ["00000", "00001"], "expressions": {"0": {"exp": "a cat", "anno_id": [3]}}}}}
    json.dump(meta, open(data_root / "mevis" / "valid_u" / "meta_expressions.json", "w"))
    rng = np.random.default_rng(1)
    for root, tail in (("grid_tracks", ("vidA",)), ("gdino_tracks", ("vidA", "0"))):
        mdir = track_root / root / "mevis" / "valid_u" / "sam2_masklets"
        tdir = track_root / root / "mevis" / "valid_u" / "sam2_object_tokens"
        for t in tail:
            mdir, tdir = mdir / t, tdir / t
        os.makedirs(mdir), os.makedirs(tdir)
        for aid in (2, 11):
            mask = (rng.uniform(size=(4, 6)) < 0.5).astype(np.uint8)
            info = {"anno_id": aid, "prompt_type": "X", "rle": [sdata.rle_encode_uncompressed(mask)] * 2, "iou": {"3": 0.1 * aid}}
            json.dump(info, open(mdir / f"{aid:05d}.json", "w"))
            np.save(tdir / f"{aid:05d}.npy", rng.standard_normal((2, 256)).astype(np.float32))
    split = {"data_name": "mevis", "data_type": "valid_u", "sam2_output_dirs": "grid_tracks,gdino_tracks", "batch_size": 1}
    ds = sdata.TrackDataset(split, str(data_root), str(track_root))
    s = ds[0]
    assert s["object_tokens"].shape == (4, 2, 256) and s["sam2_anno_id"] == [2, 11, 2, 11]
    assert s["root_type"] == ["grid_tracks", "grid_tracks", "gdino_tracks", "gdino_tracks"]
    torch.testing.assert_close(s["labels"]["iou"], torch.tensor([0.2, 1.1, 0.2, 1.1]))
    merged = ds.merged_masklet("vidA", "0", np.array([1, 0, 0, 1]))
    assert merged.shape == (2, 4, 6) and merged.dtype == bool or merged.dtype == np.uint8
    b = sdata.collate([s])
    assert b["object_tokens"].shape == (1, 4, 2, 256) and b["expression"] == ["a cat"]


def test_ragged_batcher_groups_samples_and_shares_videos():
    """Host logic of the ragged entry points: consecutive samples are grouped up to the sample / token-row budgets, the
    expressions of one video share one entry of the video list, and every sample is yielded exactly once."""
    ds = sdata.SyntheticTracks(n_samples=23, token_dim=8, seed=3, per_video=4, ragged=True)
    shapes = {v: ds.video_shape(v) for v in range(6)}
    assert len(set(shapes.values())) > 1 and all(8 <= n <= 80 and 20 <= t <= 200 for n, t in shapes.values())
    assert torch.equal(ds[4]["object_tokens"], ds[7]["object_tokens"]) and ds[4]["token_key"] == ds[7]["token_key"]
    assert ds[3]["token_key"] != ds[4]["token_key"]
    seen = []
    for batch in sdata.RaggedBatcher(ds, range(23), max_samples=10, max_rows=1 << 30):
        assert len(batch["samples"]) <= 10 and len(batch["sample_video"]) == len(batch["samples"])
        assert max(batch["sample_video"]) == len(batch["videos"]) - 1
        for smp, v in zip(batch["samples"], batch["sample_video"]):
            vid = int(smp["video_id"].split("_")[1])
            assert tuple(batch["videos"][v].shape[:2]) == shapes[vid] and "object_tokens" not in smp
            seen.append((smp["video_id"], smp["expression_id"]))
        assert len(batch["videos"]) <= (len(batch["samples"]) + 3) // 4 + 1  # four expressions per video share one entry
    assert len(seen) == 23 and len(set(seen)) == 23
    # the row budget closes a batch early; a single video larger than the budget still forms its own batch
    small = list(sdata.RaggedBatcher(ds, range(23), max_samples=1000, max_rows=1))
    assert len(small) == 6 and sum(len(b["samples"]) for b in small) == 23 and all(len(b["videos"]) == 1 for b in small)
    # samples without a token key are never merged
    class NoKey(torch.utils.data.Dataset):
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return {"object_tokens": torch.zeros(2, 3, 8), "expression": "x"}
    b = list(sdata.RaggedBatcher(NoKey(), range(3)))
    assert len(b) == 1 and b[0]["sample_video"] == [0, 1, 2]
    # worker processes read the samples ahead of the batcher: same batches, same order, same tensors
    def digest(batches):
        return [([tuple(v.shape) for v in bt["videos"]], bt["sample_video"], [s_["expression"] for s_ in bt["samples"]],
                 [float(v.double().sum()) for v in bt["videos"]]) for bt in batches]
    order = [5, 4, 17, 16, 3, 22, 9, 8, 11, 10, 0, 1, 2, 6, 7, 12]
    assert digest(sdata.RaggedBatcher(ds, order, max_samples=5, num_workers=2)) == digest(sdata.RaggedBatcher(ds, order, max_samples=5))


def test_staging_pool_bookkeeping(monkeypatch):
    """RaggedBatcher's staging (PinnedPool) without a GPU: pageable buffers stand in for page-locked ones.  Staged batches equal the
    unstaged ones; the copy of a further expression of a video the batch already holds goes straight back to the pool; buffers
    handed back after a batch serve the next pass (1-MiB size classes)."""
    import torch
    from sola_amd.data import PinnedPool, RaggedBatcher, SyntheticTracks

    monkeypatch.setattr(PinnedPool, "_alloc", staticmethod(lambda n: torch.empty(n, dtype=torch.uint8)))
    ds = SyntheticTracks(n_samples=48, token_dim=64, seed=5, with_labels=True, per_video=3, ragged=True)
    order = list(range(48))
    plain = [[v.clone() for v in b["videos"]] for b in RaggedBatcher(ds, order, 12, max_rows=1 << 62)]
    staged = RaggedBatcher(ds, order, 12, max_rows=1 << 62, num_workers=4, pin=False)
    staged.pin, staged.pool = True, PinnedPool()
    for _ in range(2):
        for b, ref in zip(staged, plain):
            assert len(b["videos"]) == len(ref) == len(b["pinned_bufs"])  # one buffer per VIDEO of the batch, not per sample
            for v, r in zip(b["videos"], ref):
                assert torch.equal(v, r)
            b["pinned_pool"].give_back(b["pinned_bufs"])
    # 48 samples were staged per pass; the second pass (and the duplicates' buffers within the first) came from the pool
    assert staged.pool.hits + staged.pool.misses == 96 and staged.pool.misses <= 48 and staged.pool.hits >= 48
    assert all(buf.numel() % (1 << 20) == 0 for lst in staged.pool._free.values() for buf in lst)
