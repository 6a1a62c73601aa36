#!/usr/bin/env python3
"""Evaluation entry point with the reference's CLI / layout (eval.py + the loss/selection part of evaluator.py):

    python eval.py --config mevis/default --eval_weight_epoch K [--eval_pred_threshold 0.5] [--synthetic true]

Reports the losses (train.py:98-113 assembly) and track-level TP/FP/FN/TN of the valid split and writes
``<eval_output_dir>/track_metrics.json``.  Mask-level J&F (evaluator.py:174-247) is outside the accelerated path.
"""
import json
import os

import torch

from sola_amd import dist as sdist
from sola_amd.config import load_configs
from sola_amd.data import make_loader
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder
from train import run_split


@torch.no_grad()
def evaluate(cfg):
    rank, local_rank, world = sdist.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    module = LanguageAlignedTrackSelectionModule(cfg["model"])
    module.load_state_dict(torch.load(cfg["eval"]["weight_path"], map_location="cpu", weights_only=True))
    module = module.to(device).eval()
    text = TextEncoder(cfg["model"]["roberta_version"], cfg["model"]["lang_token_dim"], device,
                       allow_standin=bool(cfg.get("synthetic", False)))
    loader, _ = make_loader(cfg["dataset"], "valid", rank, world, cfg.get("synthetic", None), cfg["model"])
    tcfg = dict(cfg["train"])
    tcfg["pred_threshold"] = cfg["eval"]["pred_threshold"]
    m = run_split(module, text, loader, tcfg, device, False, None, world)
    m["precision"] = m["tp"] / max(m["tp"] + m["fp"], 1.0)
    m["recall"] = m["tp"] / max(m["tp"] + m["fn"], 1.0)
    if rank == 0:
        print(json.dumps(m))
        with open(os.path.join(cfg["results"]["eval_output_dir"], "track_metrics.json"), "w") as f:
            json.dump(m, f, indent=2)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    evaluate(load_configs("eval"))
