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

    def _track_infos(self, mdir):
        """The small fields of a masklet directory's per-track JSON files (anno_id, prompt_type, iou table), parsed once: the files
        also hold every frame's RLE, and parsing them again for every expression of the video in every epoch is interpreter-bound
        work that reader THREADS cannot overlap (dataloader.py:187-199 re-reads them; the values are the same)."""
        cache = self.__dict__.setdefault("_info_cache", {})
        infos = cache.get(mdir)
        if infos is None:
            infos = []
            for name in sorted(os.listdir(mdir)):
                with open(os.path.join(mdir, name), "r") as f:
                    info = json.load(f)
                infos.append({"anno_id": info["anno_id"], "prompt_type": info["prompt_type"], "iou": info.get("iou", {})})
            cache[mdir] = infos
        return infos

    def __getitem__(self, idx):
        s = self.samples[idx]
        tokens, iou, root_type, prompt_type, sam2_ids, gt_ids = [], [], [], [], [], []
        has_gt = s["anno_ids"][0] >= 0
        for root in self.roots:
            mdir, tdir = self._dirs(root, s["video_id"], s["expression_id"])
            for info in self._track_infos(mdir):
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
        # roots with per-expression tracks (GroundingDINO prompts) make the token set expression-specific; grid / GT roots
        # are per video, so all expressions of the video score the same tokens (shared by the ragged forward)
        per_expression = any("gdino" in root for root in self.roots)
        key = (s["video_id"], s["expression_id"]) if per_expression else (s["video_id"],)
        return {**s, "object_tokens": torch.stack(tokens, 0), "labels": {"iou": torch.tensor(iou)} if has_gt else None,
                "root_type": root_type, "prompt_type": prompt_type, "sam2_anno_id": sam2_ids, "gt_anno_id": gt_ids,
                "token_key": "/".join(key)}

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
    ~20 % positives, a made-up expression string.  ``per_video`` expressions share one video (same tokens, as the grid
    tracks of dataloader.py:87-199 do: <track_root>/.../sam2_object_tokens/<video>/).  With ``ragged`` every video draws its
    own N in [8, 80] and T in [20, 200] (a MeViS-like mix); otherwise all videos are n_tracks x n_frames."""

    def __init__(self, n_samples=32, n_tracks=64, n_frames=32, token_dim=256, seed=0, with_labels=True, per_video=4, ragged=False):
        self.n, self.N, self.T, self.d, self.seed, self.with_labels = n_samples, n_tracks, n_frames, token_dim, seed, with_labels
        self.per_video, self.ragged = max(1, int(per_video)), bool(ragged)

    def __len__(self):
        return self.n

    def video_shape(self, vid):
        if not self.ragged:
            return self.N, self.T
        rng = np.random.Generator(np.random.PCG64(self.seed * 7919 + 104729 * (vid + 1)))
        return int(rng.integers(8, 81)), int(rng.integers(20, 201))

    def token_key(self, idx):
        """Samples with the same key have the same object tokens (one video's tracks): scored once per video."""
        return f"synthetic_{idx // self.per_video:04d}"

    def __getitem__(self, idx):
        vid = idx // self.per_video
        N, T = self.video_shape(vid)
        last = self.__dict__.get("_last_video")
        if last is not None and last[0] == vid:  # consecutive samples of one video: generate its tokens once
            tok = last[1]
        else:
            vrng = np.random.Generator(np.random.PCG64(self.seed * 1000003 + 7 * vid + 1))  # tokens belong to the VIDEO
            tok = torch.from_numpy(vrng.standard_normal((N, T, self.d), dtype=np.float32))  # drawn as f32: half the host time
            self._last_video = (vid, tok)
        rng = np.random.Generator(np.random.PCG64(self.seed * 1000003 + 1000 * idx + 3))
        iou = torch.from_numpy(np.where(rng.uniform(size=N) < 0.2, 0.9, 0.1).astype(np.float32))
        words = ["the", "object", "number", str(idx), "moving", "left", "behind", "a", "tree", "quickly"][: 3 + idx % 8]
        return {"video_id": f"synthetic_{vid:04d}", "expression_id": str(idx % self.per_video), "expression": " ".join(words),
                "anno_ids": [0], "frames": [f"{t:05d}" for t in range(T)], "object_tokens": tok,
                "labels": {"iou": iou} if self.with_labels else None, "root_type": ["synthetic"] * N,
                "prompt_type": ["SYNTHETIC"] * N, "sam2_anno_id": list(range(N)), "gt_anno_id": [0] * N,
                "token_key": self.token_key(idx)}


class PinnedPool:
    """Recycled page-locked staging buffers in 1-MiB size classes.  ``Tensor.pin_memory()`` per sample page-locks fresh memory every
    time (20 ms per 4 MB from one thread, tools/pin_copy_probe.py) and copies with torch's intra-op thread pool, which sixteen reader
    threads oversubscribe: they staged 860 samples/s where they drew 1 540, and more threads made it worse (tools/batcher_rate.py).
    A reader thread instead copies its sample into a buffer of the pool with one plain memcpy, and the
    uploader returns the buffers of a batch once its host-to-device copies have completed.  A buffer that is never returned is
    ordinary pinned tensor storage: it is freed with its last reference."""

    def __init__(self):
        import threading
        self._free, self._lock = {}, threading.Lock()
        self.hits = self.misses = 0  # stage() calls served from the pool / by page-locking fresh memory

    @staticmethod
    def _alloc(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)

    def stage(self, t):
        """Copy of ``t`` in page-locked memory + the pool buffer that holds it."""
        nb = t.numel() * t.element_size()
        cls = max(1, (nb + (1 << 20) - 1) >> 20)
        with self._lock:
            lst = self._free.get(cls)
            buf = lst.pop() if lst else None
            self.hits += buf is not None
            self.misses += buf is None
        if buf is None:
            buf = self._alloc(cls << 20)
        dst = buf[:nb].view(t.dtype).view(t.shape)
        if t.is_contiguous() and t.dtype in (torch.float32, torch.float16, torch.uint8, torch.int32, torch.int64):
            np.copyto(dst.numpy(), t.numpy())  # one memcpy outside the interpreter lock.  NOT Tensor.copy_: from sixteen reader threads
            # its intra-op thread pool is oversubscribed sixteen-fold (tools/pin_copy_probe.py: 13 GB/s aggregate against 21 from one
            # thread; the batcher staged 770 samples/s with copy_, 1 680 with OMP_NUM_THREADS=1 or this)
        else:
            dst.copy_(t)
        return dst, buf

    def give_back(self, bufs):
        with self._lock:
            for b in bufs:
                self._free.setdefault(b.numel() >> 20, []).append(b)


class RaggedBatcher:
    """Groups consecutive samples of a dataset shard into ragged batches for ``forward_ragged``: up to ``max_samples``
    (video, expression) pairs per launch, bounded by ``max_rows`` object-token rows (sum of N*T over the batch's videos,
    which bounds the workspace), with samples of one video (same ``token_key``) sharing one entry of the video list.

    Yields dictionaries: ``videos`` (list of [N_v,T_v,d] tensors), ``sample_video`` (index per sample), ``samples`` (the
    dataset's sample dictionaries, tokens removed)."""

    def __init__(self, dataset, indices, max_samples=128, max_rows=1 << 20, num_workers=0, pin=False):
        self.dataset, self.indices = dataset, list(indices)
        self.max_samples, self.max_rows = int(max_samples), int(max_rows)
        # pin: the reader threads also move a sample's tokens into page-locked memory (Tensor.pin_memory: ~2.5 ms per 2 MB sample,
        # which held the pipeline at 400 samples/s when the ONE upload thread of DevicePrefetcher did it for every sample)
        self.pin = bool(pin) and torch.cuda.is_available()
        self.pool = PinnedPool() if self.pin else None  # batches carry their staging buffers in "pinned_bufs" (DevicePrefetcher returns them)
        # reader THREADS that fetch (or, for the synthetic stand-in, draw) the samples ahead of the batcher, in order.  Threads, not
        # worker processes: a sample is 1-16 MB of object tokens and a tensor that crosses a process boundary through shared
        # memory costs ~5 ms to touch on the receiving side (0.4 GB/s: tools/train_pipeline_profile.py measured 330 samples/s
        # behind 32 worker processes that produced 1 000) - file reads and numpy's generators release the interpreter lock
        self.num_workers = int(num_workers)

    def set_indices(self, indices):
        """Another pass over the same dataset in another order (the next epoch's shuffle)."""
        self.indices = list(indices)

    def _samples(self):
        if self.num_workers <= 0 or len(self.indices) < 2:
            for idx in self.indices:
                yield idx, self.dataset[idx]
            return
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        def fetch(idx):
            smp = self.dataset[idx]
            if self.pin and isinstance(smp.get("object_tokens"), torch.Tensor) and not smp["object_tokens"].is_cuda:
                smp["object_tokens"], smp["_pin_buf"] = self.pool.stage(smp["object_tokens"])
            return smp

        ahead = 4 * self.num_workers
        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            pending = deque()
            it = iter(self.indices)
            for idx in it:
                pending.append((idx, pool.submit(fetch, idx)))
                if len(pending) >= ahead:
                    break
            while pending:
                idx, fut = pending.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append((nxt, pool.submit(fetch, nxt)))
                yield idx, fut.result()

    def __iter__(self):
        videos, keys, sample_video, samples, rows, bufs = [], {}, [], [], 0, []

        def batch():
            b = {"videos": videos, "sample_video": sample_video, "samples": samples}
            if self.pool is not None:
                b["pinned_bufs"], b["pinned_pool"] = bufs, self.pool
            return b

        for idx, smp in self._samples():
            tok = smp.pop("object_tokens")
            buf = smp.pop("_pin_buf", None)
            key = smp.get("token_key", ("sample", idx))
            new_rows = 0 if key in keys else int(tok.shape[0]) * int(tok.shape[1])
            if samples and (len(samples) >= self.max_samples or (new_rows > 0 and rows + new_rows > self.max_rows)):
                yield batch()
                videos, keys, sample_video, samples, rows, bufs = [], {}, [], [], 0, []
                new_rows = int(tok.shape[0]) * int(tok.shape[1])
            if key not in keys:
                keys[key] = len(videos)
                videos.append(tok)
                rows += new_rows
                if buf is not None:
                    bufs.append(buf)
            elif buf is not None:
                self.pool.give_back([buf])  # a further expression of a video the batch already holds: its copy is not used
            sample_video.append(keys[key])
            samples.append(smp)
        if samples:
            yield batch()


class DevicePrefetcher:
    """Iterates ragged batches with their ``videos`` already on ``device``: a background thread pulls the next batch from the
    dataset, pins its object tokens and issues the host->device copies on a side stream while the caller scores the current
    batch (the copies of a 128-sample batch are ~160 MB: 10-15 ms from pageable memory on the compute stream).  The consumer's
    stream waits for the copy event before it touches the tensors."""

    def __init__(self, batches, device, depth=2):
        self.batches, self.device, self.depth = batches, device, int(depth)

    def __iter__(self):
        import queue
        import threading

        import torch as _t

        q = queue.Queue(maxsize=self.depth)
        stream = _t.cuda.Stream(device=self.device)
        done = object()
        stop = threading.Event()

        def put(item):  # never blocks for good: a consumer that left (break / exception) sets `stop`
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                _t.cuda.set_device(self.device)
                inflight = []  # (copy event, staging buffers, pool) of uploaded batches: buffers go back once the copies are done
                for b in self.batches:
                    if stop.is_set():
                        return
                    while inflight and inflight[0][0].query():
                        _ev, bufs, pool = inflight.pop(0)
                        pool.give_back(bufs)
                    b = dict(b)
                    bufs, pool = b.pop("pinned_bufs", None), b.pop("pinned_pool", None)
                    with _t.cuda.stream(stream):
                        vids = []
                        for v in b["videos"]:  # reader threads hand over page-locked tensors (RaggedBatcher pin=True): only the copy is left
                            v = v if (v.is_cuda or v.is_pinned()) else v.pin_memory()
                            vids.append(v.to(self.device, non_blocking=True))
                        ev = _t.cuda.Event()
                        ev.record(stream)
                    if bufs and pool is not None:
                        inflight.append((ev, bufs, pool))
                    if not put((dict(b, videos=vids), ev)):
                        return
                put(done)
            except BaseException as e:  # surface loader errors in the consumer
                put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is done:
                    break
                if isinstance(item, BaseException):
                    raise item
                b, ev = item
                _t.cuda.current_stream(self.device).wait_event(ev)
                for v in b["videos"]:
                    v.record_stream(_t.cuda.current_stream(self.device))
                yield b
        finally:
            # the consumer may leave early (break, an exception in its loop body, generator close): release the worker, drop the
            # batches it already uploaded (a 128-sample batch is ~160 MB of device + pinned memory) and join it
            stop.set()
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            th.join(timeout=30)


def make_ragged_batches(cfg_dataset: dict, split: str, rank=0, world=1, synthetic=None, model_cfg=None):
    """(iterable of ragged batches over the rank's shard, dataset).  ``dataset.ragged_max_samples`` / ``ragged_max_rows``
    bound a batch (defaults 128 samples, 2^20 token rows = 1 GiB of object tokens)."""
    from .dist import shard_indices

    ds = make_dataset(cfg_dataset, split, synthetic, model_cfg)
    # contiguous shards keep a video's expressions on one rank (round-robin would scatter them and lose the sharing)
    n = len(ds)
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    idx = list(range(lo, hi)) if world > 1 else shard_indices(n, 0, 1)
    # reader threads (dataset.reader_threads, default 16): the expressions of a video are consecutive and share its tokens, so
    # concurrent readers fetch them once per expression where one reader's one-entry cache fetched them once per video - with four
    # threads that cancels out (inference.py: 446 -> 380 samples/s wall), with sixteen it is 2.4x (420 -> 1 020 samples/s wall)
    readers = int(cfg_dataset.get("reader_threads", min(16, os.cpu_count() or 1)))
    return RaggedBatcher(ds, idx, int(cfg_dataset.get("ragged_max_samples", 128)), int(cfg_dataset.get("ragged_max_rows", 1 << 20)),
                         num_workers=readers, pin=True), ds


def make_ragged_train_batches(cfg_dataset: dict, rank=0, world=1, synthetic=None, model_cfg=None, samples_per_step=64, epoch=0, reuse=None):
    """Ragged TRAINING batches of one epoch for this rank: the train split shuffled with a seed every rank derives from the
    epoch, sharded i % world == rank with the short shards padded (every rank must issue the same number of optimizer steps:
    each ends in one gradient all-reduce), cut into batches of exactly ``samples_per_step`` samples (the last one shorter) -
    bounded by the sample count only, so all ranks see the same number of batches."""
    from .dist import shard_indices

    ds = reuse.dataset if reuse is not None else make_dataset(cfg_dataset, "train", synthetic, model_cfg)
    n = len(ds)
    order = np.random.Generator(np.random.PCG64(1234567 + int(epoch))).permutation(n).tolist()
    mine = [order[i] for i in shard_indices(n, rank, world, pad=True)]
    if reuse is not None:  # the next epoch of the same batcher: its worker processes stay
        reuse.set_indices(mine)
        return reuse, ds
    # dataset.reader_threads (default: 16, at most the host's cores): threads that read / draw the samples ahead of the batcher and
    # page-lock their tokens; dataset.num_workers (the reference's DataLoader key) still sizes the one-sample-per-step loader
    readers = int(cfg_dataset.get("reader_threads", min(16, os.cpu_count() or 1)))
    return RaggedBatcher(ds, mine, int(samples_per_step), max_rows=1 << 62, num_workers=readers, pin=True), ds


def collate(batch):
    out = {k: [b[k] for b in batch] for k in batch[0] if k not in ("object_tokens", "labels", "token_key")}
    out["object_tokens"] = torch.stack([b["object_tokens"] for b in batch], 0)
    out["labels"] = None if batch[0]["labels"] is None else {"iou": torch.stack([b["labels"]["iou"] for b in batch], 0)}
    return out


def make_dataset(cfg_dataset: dict, split: str, synthetic=None, model_cfg=None):
    sc = cfg_dataset[split]
    use_syn = bool(synthetic)
    if not use_syn and not os.path.isdir(str(cfg_dataset.get("track_root", ""))):
        raise FileNotFoundError(f"dataset.track_root '{cfg_dataset.get('track_root')}' does not exist; pass --synthetic true "
                                "for a plumbing run on generated tracks")
    if use_syn:
        return SyntheticTracks(n_samples=int(cfg_dataset.get("synthetic_samples", 32)), n_tracks=int(cfg_dataset.get("synthetic_tracks", 64)),
                               n_frames=int(cfg_dataset.get("synthetic_frames", 32)),
                               token_dim=(model_cfg or {}).get("object_token_dim", 256), seed={"train": 0, "valid": 1, "test": 2}[split],
                               with_labels=split != "test", per_video=int(cfg_dataset.get("synthetic_per_video", 4)),
                               ragged=bool(cfg_dataset.get("synthetic_ragged", False)))
    return TrackDataset(sc, cfg_dataset["data_root"], cfg_dataset["track_root"])


def make_loader(cfg_dataset: dict, split: str, rank=0, world=1, synthetic=None, model_cfg=None):
    """DataLoader over the rank's shard (sample i belongs to rank i % world)."""
    from .dist import shard_indices

    sc = cfg_dataset[split]
    use_syn = bool(synthetic)
    ds = make_dataset(cfg_dataset, split, synthetic, model_cfg)
    # training shards are padded to equal length: one gradient all-reduce per step must meet its peers on every rank
    sub = torch.utils.data.Subset(ds, shard_indices(len(ds), rank, world, pad=(split == "train")))
    # worker processes also for the synthetic stand-in: drawing a sample's tokens on the host (numpy, 0.5 M normals at 64 x 32
    # tracks x frames) is ~5 ms, 40x the GPU time of the sample at 64 samples per step (tools/train_rate.sh)
    readers = int(cfg_dataset.get("reader_threads", min(16, os.cpu_count() or 1)))
    if readers > 0:
        # reader THREADS + page-locked batches (RaggedBatcher's note: tensors that cross a process boundary are ~5 ms per 2 MB
        # on the receiving side; train.py at one sample per step: 250 samples/s behind four worker processes, step-bound 330-350
        # behind threads).  dataset.reader_threads 0 = torch's DataLoader with dataset.num_workers processes, as the reference.
        return ThreadLoader(sub, int(sc.get("batch_size", 1)), split == "train", readers), ds
    nw = int(cfg_dataset.get("num_workers", 0))
    loader = torch.utils.data.DataLoader(sub, batch_size=sc.get("batch_size", 1), shuffle=(split == "train"), num_workers=nw,
                                         pin_memory=True, collate_fn=collate, persistent_workers=nw > 0,
                                         prefetch_factor=4 if nw > 0 else None)
    return loader, ds


class ThreadLoader:
    """DataLoader(sub, batch_size, shuffle, collate_fn=collate, pin_memory=True) on a thread pool: the samples of an epoch are
    fetched in order by ``threads`` readers running ahead of the consumer, collated per batch and page-locked.  A new shuffle
    (torch's global generator, like DataLoader's RandomSampler) on every pass."""

    def __init__(self, dataset, batch_size=1, shuffle=False, threads=16):
        self.dataset, self.batch_size, self.shuffle, self.threads = dataset, max(1, int(batch_size)), bool(shuffle), int(threads)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        groups = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        pin = torch.cuda.is_available()

        def fetch(group):
            b = collate([self.dataset[i] for i in group])
            if pin:
                b["object_tokens"] = b["object_tokens"].pin_memory()
                if b["labels"] is not None:
                    b["labels"] = {k: v.pin_memory() for k, v in b["labels"].items()}
            return b

        ahead = 4 * self.threads
        with ThreadPoolExecutor(max_workers=self.threads) as pool:
            pending = deque()
            it = iter(groups)
            for g in it:
                pending.append(pool.submit(fetch, g))
                if len(pending) >= ahead:
                    break
            while pending:
                fut = pending.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(fetch, nxt))
                yield fut.result()
