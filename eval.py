#!/usr/bin/env python3
"""Evaluation entry point with the reference's CLI / layout (eval.py + the loss/selection part of evaluator.py):

    python eval.py --config mevis/default --eval_weight_epoch K [--eval_pred_threshold 0.5] [--synthetic true]

Reports the losses and track-level TP/FP/FN/TN of the valid split and writes ``<eval_output_dir>/track_metrics.json``.
Scoring runs on ragged batches (sola_forward_ragged: up to ``--ragged_max_samples`` samples of different shapes per launch,
one pass of the text-independent half per video); every sample's losses are its own means, as at the reference's batch
size of 1.  Mask-level J&F (evaluator.py:174-247) is outside the accelerated path.

BCE convention: ``bce`` / ``total`` follow train.py:98-113 (BCE-with-logits on the LOGITS, what the
network is trained and validated with).  The reference's evaluator applies binary_cross_entropy_with_logits to the
already SIGMOID-ed scores (evaluator.py:101,107-111) - a double sigmoid; that number is reported separately as
``bce_evaluator_convention`` so eval JSONs can be compared line by line, and never enters ``total``.
"""
import json
import os

import torch

from sola_amd import dist as sdist
from sola_amd.config import load_configs
from sola_amd.data import make_ragged_batches
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder
from train import run_split_ragged


@torch.no_grad()
def evaluate(cfg):
    rank, local_rank, world = sdist.init_from_env()
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    module = LanguageAlignedTrackSelectionModule(cfg["model"])
    module.load_state_dict(torch.load(cfg["eval"]["weight_path"], map_location="cpu", weights_only=True))
    module = module.to(device).eval()
    text = TextEncoder(cfg["model"]["roberta_version"], cfg["model"]["lang_token_dim"], device,
                       allow_standin=bool(cfg.get("synthetic", False)))
    batches, _ = make_ragged_batches(cfg["dataset"], "valid", rank, world, cfg.get("synthetic", None), cfg["model"])
    tcfg = dict(cfg["train"])
    tcfg["pred_threshold"] = cfg["eval"]["pred_threshold"]
    m = run_split_ragged(module, text, batches, tcfg, device, world)
    m["text_encoder"] = text.kind
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
