#!/usr/bin/env python3
"""Inference entry point with the reference's CLI / layout (inference.py of cvlab-kaist/SOLA):

    python inference.py --config mevis/default --eval_weight_epoch K [--eval_pred_threshold 0.5] [--synthetic true]

Loads ``<output_dir>/<exp_name>/<train.data_name>/epoch_K.pth`` (a reference checkpoint loads unchanged: same 84
state_dict keys), scores every (video, expression) of the test split with sola_forward + sola_select on the GPU, and
writes ``<test_output_dir>/.../<video>/<expression>/<frame>.png`` by OR-merging the selected tracks' masklets.
With several GPUs (torchrun) the samples are sharded i % world == rank; no collective is involved.
"""
import os

import numpy as np
import torch

from sola_amd import dist as sdist
from sola_amd import ops
from sola_amd.config import load_configs
from sola_amd.data import make_loader
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder


@torch.no_grad()
def inference(cfg):
    rank, local_rank, world = sdist.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    module = LanguageAlignedTrackSelectionModule(cfg["model"])
    module.load_state_dict(torch.load(cfg["eval"]["weight_path"], map_location="cpu", weights_only=True))
    module = module.to(device).eval()
    text = TextEncoder(cfg["model"]["roberta_version"], cfg["model"]["lang_token_dim"], device,
                       allow_standin=bool(cfg.get("synthetic", False)))
    loader, dataset = make_loader(cfg["dataset"], "test", rank, world, cfg.get("synthetic", None), cfg["model"])
    thr = cfg["eval"]["pred_threshold"]
    out_dir = cfg["results"]["test_output_dir"]
    n_selected = n_tracks = 0
    for batch in loader:
        obj = batch["object_tokens"].to(device, non_blocking=True)
        lang, _pos = text.encode(batch["expression"])
        score, _ = module(obj, lang)
        prob, pred = ops.select(score, thr)  # inference.py:59-60
        pred = pred.cpu().numpy()
        for b in range(pred.shape[0]):
            vid, eid = batch["video_id"][b], batch["expression_id"][b]
            n_selected += int(pred[b].sum())
            n_tracks += pred.shape[1]
            if hasattr(dataset, "merged_masklet"):
                from PIL import Image

                masklet = dataset.merged_masklet(vid, eid, pred[b], device=device).cpu().numpy()  # RLE decode + OR on the GPU
                os.makedirs(os.path.join(out_dir, vid, eid), exist_ok=True)
                for frame_id, mask in zip(batch["frames"][b], masklet):
                    Image.fromarray((np.asarray(mask) * 255).astype(np.uint8)).save(os.path.join(out_dir, vid, eid, f"{frame_id}.png"))
            else:  # synthetic tracks have no masklets: keep the decision vector
                os.makedirs(os.path.join(out_dir, vid), exist_ok=True)
                np.save(os.path.join(out_dir, vid, f"{eid}_pred.npy"), pred[b])
    print(f"[rank {rank}] selected {n_selected} of {n_tracks} tracks; outputs in {out_dir}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    inference(load_configs("inference"))
