"""Host-side data plumbing for the entry points (out of the accelerated scope; it only has to honour the on-disk
contract between track generation and track selection, SURVEY §5 / dataloader.py:87-199):

  <track_root>/<root>/<data_name>/<data_type>/sam2_object_tokens/<video>[/<expression>]/<anno_id:05d>.npy   [T,256] fp32
  <track_root>/<root>/<data_name>/<data_type>/sam2_masklets/<video>[/<expression>]/<anno_id:05d>.json
        {"anno_id", "rle": [{size:[h,w], counts:str} x T], "prompt_type", optional "iou"/"precision"/"recall": {gt_id: float}}

(the <expression> level exists only for roots whose name contains "gdino").  ``SyntheticTracks`` stands in when no
dataset is mounted (this image has none), with the same sample dictionary.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

NO_OBJECT_ID = -1


# ------------------------------------------------------------------------------------------- COCO RLE (no pycocotools)
def rle_counts_from_string(s: str):
    """COCO compressed-RLE string -> run lengths (6 bits per char, sign-extended, delta-coded from the 3rd run on)."""
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(rle: dict) -> np.ndarray:
    """{size:[h,w], counts: str | list} -> uint8 [h,w] (runs are column-major, starting with a run of zeros)."""
    h, w = rle["size"]
    counts = rle["counts"]
    if isinstance(counts, (bytes, bytearray)):
        counts = counts.decode("utf-8")
    if isinstance(counts, str):
        counts = rle_counts_from_string(counts)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T


def rle_encode_uncompressed(mask: np.ndarray) -> dict:
    """uint8 [h,w] -> {size, counts: list} (used by the synthetic writer and tests)."""
    h, w = mask.shape
    flat = np.asarray(mask, dtype=np.uint8).T.reshape(-1)
    change = np.flatnonzero(np.diff(flat)) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(bounds).tolist()
    if flat.size and flat[0] == 1:
        runs = [0] + runs
    return {"size": [h, w], "counts": runs}


def masklet_decode(rle_list) -> np.ndarray:
    """List of per-frame RLE dicts (None for missing frames) -> uint8 [T,h,w] (dataloader.py:353-369)."""
    frames, h, w = [], 0, 0
    for r in rle_list:
        if isinstance(r, dict):
            m = rle_decode(r)
            h, w = m.shape
            frames.append(m)
        else:
            frames.append(None)
    return np.stack([f if f is not None else np.zeros((h, w), dtype=np.uint8) for f in frames], axis=0)


# ------------------------------------------------------------------------------------------- datasets
class TrackDataset(torch.utils.data.Dataset):
    """Reads precomputed SAM2 tracks for one split (same sample dictionary as dataloader.py:187-199)."""

    def __init__(self, split_cfg: dict, data_root: str, track_root: str):
        self.data_name, self.data_type = split_cfg["data_name"], split_cfg["data_type"]
        self.data_root, self.track_root = data_root, track_root
        self.roots = split_cfg["sam2_output_dirs"].split(",")
        if self.data_name == "mevis":
            meta_path = os.path.join(data_root, self.data_name, self.data_type, "meta_expressions.json")
        elif self.data_name in ("ref-ytbvos", "ref-davis"):
            meta_path = os.path.join(data_root, self.data_name, "meta_expressions", self.data_type, "meta_expressions.json")
        else:
            raise ValueError(f"Invalid data_name: {self.data_name}")
        with open(meta_path, "r") as f:
            self.meta = json.load(f)
        self.samples = []
        for vid, vmeta in self.meta["videos"].items():
            for eid, emeta in vmeta["expressions"].items():
                if self.data_name == "mevis":
                    anno_ids = emeta.get("anno_id", [NO_OBJECT_ID])
                else:
                    anno_ids = [int(emeta.get("obj_id", NO_OBJECT_ID))]
                self.samples.append({"video_id": vid, "expression_id": eid, "expression": emeta["exp"], "anno_ids": anno_ids,
                                     "frames": vmeta["frames"]})

    def __len__(self):
        return len(self.samples)

    def _dirs(self, root, video_id, expression_id):
        base = os.path.join(self.track_root, root, self.data_name, self.data_type)
        tail = (video_id, expression_id) if "gdino" in root else (video_id,)
        return os.path.join(base, "sam2_masklets", *tail), os.path.join(base, "sam2_object_tokens", *tail)

    def __getitem__(self, idx):
        s = self.samples[idx]
        tokens, iou, root_type, prompt_type, sam2_ids, gt_ids = [], [], [], [], [], []
        has_gt = s["anno_ids"][0] >= 0
        for root in self.roots:
            mdir, tdir = self._dirs(root, s["video_id"], s["expression_id"])
            for name in sorted(os.listdir(mdir)):
                with open(os.path.join(mdir, name), "r") as f:
                    info = json.load(f)
                best, best_id = 0.0, NO_OBJECT_ID
                if has_gt:
                    for a in s["anno_ids"]:
                        v = info.get("iou", {}).get(str(a), 0.0)
                        if v > best:
                            best, best_id = v, a
                iou.append(best)
                gt_ids.append(best_id)
                sam2_ids.append(info["anno_id"])
                root_type.append(os.path.basename(root))
                prompt_type.append(info["prompt_type"])
                tokens.append(torch.from_numpy(np.load(os.path.join(tdir, f"{info['anno_id']:05d}.npy"))).float())
        assert tokens, "object_tokens is empty"
        return {**s, "object_tokens": torch.stack(tokens, 0), "labels": {"iou": torch.tensor(iou)} if has_gt else None,
                "root_type": root_type, "prompt_type": prompt_type, "sam2_anno_id": sam2_ids, "gt_anno_id": gt_ids}

    def merged_masklet(self, video_id, expression_id, preds, device=None):
        """OR of the RLE-decoded masklets of the selected tracks (dataloader.py:305-351).  With ``device`` the run
        lengths of all selected tracks are decoded and OR-ed in one launch (seg_utils.rle_merge_or) and the result is a
        uint8 CUDA tensor; without it the host decoder is used (numpy array)."""
        merged, i, selected = None, 0, []
        for root in self.roots:
            mdir, _ = self._dirs(root, video_id, expression_id)
            for name in sorted(os.listdir(mdir)):
                with open(os.path.join(mdir, name), "r") as f:
                    info = json.load(f)
                if preds[i] > 0 and device is not None:
                    selected.append(info["rle"])
                elif preds[i] > 0:
                    m = masklet_decode(info["rle"])
                    merged = m if merged is None else np.logical_or(merged, m)
                elif merged is None:
                    h, w = info["rle"][0]["size"]
                    merged = np.zeros((len(info["rle"]), h, w), dtype=np.uint8)
                i += 1
        if selected:
            from . import seg_utils
            return seg_utils.rle_merge_or(selected, device)
        if merged is not None and device is not None:
            return torch.from_numpy(np.ascontiguousarray(merged, dtype=np.uint8)).to(device)
        return merged


class SyntheticTracks(torch.utils.data.Dataset):
    """Deterministic stand-in with the real sample dictionary: N tracks x T frames of N(0,1) tokens, IoU labels with
    ~20 % positives, a made-up expression string."""

    def __init__(self, n_samples=32, n_tracks=64, n_frames=32, token_dim=256, seed=0, with_labels=True):
        self.n, self.N, self.T, self.d, self.seed, self.with_labels = n_samples, n_tracks, n_frames, token_dim, seed, with_labels

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.Generator(np.random.PCG64(self.seed * 1000003 + idx))
        tok = torch.from_numpy(rng.standard_normal((self.N, self.T, self.d)).astype(np.float32))
        iou = torch.from_numpy(np.where(rng.uniform(size=self.N) < 0.2, 0.9, 0.1).astype(np.float32))
        return {"video_id": f"synthetic_{idx // 4:04d}", "expression_id": str(idx % 4), "expression": f"the object number {idx} moving left",
                "anno_ids": [0], "frames": [f"{t:05d}" for t in range(self.T)], "object_tokens": tok,
                "labels": {"iou": iou} if self.with_labels else None, "root_type": ["synthetic"] * self.N,
                "prompt_type": ["SYNTHETIC"] * self.N, "sam2_anno_id": list(range(self.N)), "gt_anno_id": [0] * self.N}


def collate(batch):
    out = {k: [b[k] for b in batch] for k in batch[0] if k not in ("object_tokens", "labels")}
    out["object_tokens"] = torch.stack([b["object_tokens"] for b in batch], 0)
    out["labels"] = None if batch[0]["labels"] is None else {"iou": torch.stack([b["labels"]["iou"] for b in batch], 0)}
    return out


def make_loader(cfg_dataset: dict, split: str, rank=0, world=1, synthetic=None, model_cfg=None):
    """DataLoader over the rank's shard (sample i belongs to rank i % world)."""
    from .dist import shard_indices

    sc = cfg_dataset[split]
    use_syn = bool(synthetic)
    if not use_syn and not os.path.isdir(str(cfg_dataset.get("track_root", ""))):
        raise FileNotFoundError(f"dataset.track_root '{cfg_dataset.get('track_root')}' does not exist; pass --synthetic true "
                                "for a plumbing run on generated tracks")
    if use_syn:
        ds = SyntheticTracks(n_samples=int(cfg_dataset.get("synthetic_samples", 32)), n_tracks=int(cfg_dataset.get("synthetic_tracks", 64)),
                             n_frames=int(cfg_dataset.get("synthetic_frames", 32)),
                             token_dim=(model_cfg or {}).get("object_token_dim", 256), seed={"train": 0, "valid": 1, "test": 2}[split],
                             with_labels=split != "test")
    else:
        ds = TrackDataset(sc, cfg_dataset["data_root"], cfg_dataset["track_root"])
    # training shards are padded to equal length: one gradient all-reduce per step must meet its peers on every rank
    sub = torch.utils.data.Subset(ds, shard_indices(len(ds), rank, world, pad=(split == "train")))
    loader = torch.utils.data.DataLoader(sub, batch_size=sc.get("batch_size", 1), shuffle=(split == "train"),
                                         num_workers=0 if use_syn else int(cfg_dataset.get("num_workers", 0)), pin_memory=True, collate_fn=collate)
    return loader, ds
