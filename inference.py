#!/usr/bin/env python3
"""Inference entry point with the reference's CLI / layout (inference.py of cvlab-kaist/SOLA):

    python inference.py --config mevis/default --eval_weight_epoch K [--eval_pred_threshold 0.5] [--synthetic true]
                        [--ragged_max_samples 128] [--ragged_max_rows 1048576]

Loads ``<output_dir>/<exp_name>/<train.data_name>/epoch_K.pth`` (a reference checkpoint loads unchanged: same 84
state_dict keys) and scores the test split.  The reference scores ONE (video, expression) per forward (inference.py:44-58,
batch_size 1): here up to ``ragged_max_samples`` samples of different (N, T, L) go through sola_forward_ragged in one pass,
and the expressions of one video share the text-independent half of the network (encoder + layer 0's inter-object and
motion sub-blocks).  sola_select applies the threshold (inference.py:59-60); the selected tracks' masklets are RLE-decoded
and OR-merged on the GPU and written as ``<test_output_dir>/.../<video>/<expression>/<frame>.png``.
With several GPUs (torchrun) every rank takes a contiguous block of the samples; no collective is involved.
"""
import os
import time

import numpy as np
import torch

from sola_amd import dist as sdist
from sola_amd import ops
from sola_amd.config import load_configs
from sola_amd.data import DevicePrefetcher, make_ragged_batches
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder


@torch.no_grad()
def inference(cfg):
    rank, local_rank, world = sdist.init_from_env()
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    module = LanguageAlignedTrackSelectionModule(cfg["model"])
    module.load_state_dict(torch.load(cfg["eval"]["weight_path"], map_location="cpu", weights_only=True))
    module = module.to(device).eval()
    text = TextEncoder(cfg["model"]["roberta_version"], cfg["model"]["lang_token_dim"], device,
                       allow_standin=bool(cfg.get("synthetic", False)))
    batches, dataset = make_ragged_batches(cfg["dataset"], "test", rank, world, cfg.get("synthetic", None), cfg["model"])
    thr = cfg["eval"]["pred_threshold"]
    out_dir = cfg["results"]["test_output_dir"]
    n_selected = n_tracks = n_samples = n_videos = n_calls = 0
    t_score = 0.0
    t_wall0 = time.perf_counter()
    for batch in DevicePrefetcher(batches, device):  # the next batch's tokens are copied to the GPU while this one is scored
        t0 = time.perf_counter()
        videos = batch["videos"]
        texts, _pos = text.encode_ragged([s["expression"] for s in batch["samples"]])
        module.forward_ragged(videos, texts, batch["sample_video"])
        flat, _tok, _offs, counts = module.last_ragged
        _prob, pred = ops.select(flat, thr)  # inference.py:59-60
        pred = pred.cpu().numpy()
        t_score += time.perf_counter() - t0
        n_calls += 1
        n_videos += len(videos)
        o = 0
        for smp, n in zip(batch["samples"], counts):
            p = pred[o:o + n]
            o += n
            vid, eid = smp["video_id"], smp["expression_id"]
            n_selected += int(p.sum())
            n_tracks += n
            n_samples += 1
            if hasattr(dataset, "merged_masklet"):
                from PIL import Image

                masklet = dataset.merged_masklet(vid, eid, p, device=device).cpu().numpy()  # RLE decode + OR on the GPU
                os.makedirs(os.path.join(out_dir, vid, eid), exist_ok=True)
                for frame_id, mask in zip(smp["frames"], masklet):
                    Image.fromarray((np.asarray(mask) * 255).astype(np.uint8)).save(os.path.join(out_dir, vid, eid, f"{frame_id}.png"))
            else:  # synthetic tracks have no masklets: keep the decision vector
                os.makedirs(os.path.join(out_dir, vid), exist_ok=True)
                np.save(os.path.join(out_dir, vid, f"{eid}_pred.npy"), p)
    rate = n_samples / t_score if t_score > 0 else 0.0
    wall = n_samples / max(time.perf_counter() - t_wall0, 1e-9)
    print(f"[rank {rank}] selected {n_selected} of {n_tracks} tracks over {n_samples} samples / {n_videos} video passes in {n_calls} "
          f"ragged calls ({rate:.1f} samples/s scoring incl. text encoding and the decision copy, token upload prefetched; "
          f"{wall:.1f} samples/s wall incl. dataset reads and output files; text encoder: {text.kind}); outputs in {out_dir}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    inference(load_configs("inference"))
