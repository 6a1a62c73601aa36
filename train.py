#!/usr/bin/env python3
"""Training entry point with the reference's CLI / YAML / checkpoint layout (train.py of cvlab-kaist/SOLA), running
the track-selection network, its losses, backward, gradient norms and clipping in libsola_hip.so.

    python train.py --config mevis/default [--n_epochs_override 1] [--synthetic true] [--samples_per_step 64]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config mevis/default

``--samples_per_step K`` (K > 1) trains on ragged batches: K (video, expression) samples of DIFFERENT shapes per optimizer step,
the step's objective the mean of the per-sample losses (run_train_ragged).  The default, 1, is the reference's loop
(configs/mevis/default.yaml:37): one sample per step.  Averaging K gradients before one AdamW update is not K batch-1 updates:
re-tune ``train.lr`` for K > 1 (AdamW's update size is set by lr, not by the gradient scale, so K-fold fewer updates per epoch at
the reference's lr train K-fold slower per epoch).

One process per GPU; samples are sharded i % world == rank; the only collective is the RCCL all-reduce (average) of
the 32.98M-element gradient before get_grad_norm_dict()/clipping, so every rank clips and steps identically
(effective batch = world size; the reference itself is single-process, batch 1).  AdamW / ReduceLROnPlateau stay on
PyTorch as in the reference (train.py:44-57).  Weights are saved as ``<output_dir>/epoch_{k}.pth`` (train.py:246).
"""
import math
import os
import random
import time

import numpy as np
import torch

from sola_amd import dist as sdist
from sola_amd._lib import SolaError
from sola_amd.config import load_configs
from sola_amd.data import make_loader, make_ragged_batches, make_ragged_train_batches, DevicePrefetcher
from sola_amd.loss import track_selection_losses, track_selection_losses_ragged
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def run_split(module, text, loader, tcfg, device, train, optimizer=None, world=1):
    pw, temp, aw = tcfg["positive_weight"], tcfg["temperature"], tcfg["alignment_weight"]
    module.train(train)
    sums = torch.zeros(3, device=device)
    counts = torch.zeros(4, device=device)  # TP FP FN TN
    n = n_smp = 0
    t_loop = time.time()
    for batch in loader:
        obj = batch["object_tokens"].to(device, non_blocking=True)
        n_smp += int(obj.shape[0])
        labels = (batch["labels"][tcfg["positive_metric"]] > tcfg["positive_threshold"]).float().to(device)
        lang, pos = text.encode(batch["expression"])
        if train and world == 1:
            # round 5: forward, both losses on the module's own negative tokens (train.py:92), loss.backward(), the gradient norms and the
            # clipping (train.py:95-122) as ONE library call (module.train_step -> sola_train_step); bit-identical to the statements below
            # (clipping + the AdamW update: one more launch with torch's fused arithmetic on the optimizer's own state tensors - sola_adamw_step)
            loss3, score, tokens = module.train_step(obj, lang, labels, pos, pw, temp, aw, max_grad_norm=max(float(tcfg["grad_clip_norm"]), 0.0),
                                                     optimizer=optimizer, write_back_grads=False)  # .grad is not read behind the step
            sums += loss3.detach()
            n += 1
            continue
        with torch.set_grad_enabled(train):
            score, tokens = module(obj, lang)
            # train.py:92 (batch_size = lang_tokens.shape[0])
            neg = module.negative_token.weight.clone().unsqueeze(0).repeat(lang.shape[0], 1, 1)
            loss3 = track_selection_losses(score, tokens, labels, pos, neg, pw, temp, aw)
        if train:
            optimizer.zero_grad(set_to_none=True)
            loss3[0].backward()
            sdist.allreduce_gradient_arena(module, world)  # in place on the flat arena, bucket by bucket, overlapping the backward
            # train.py:120-125 (norm, clip if the total exceeds the threshold, AdamW): the norm is reduced and the decision taken on the
            # device (the reference does 83 .item() calls here), and clip + update are ONE launch with torch's fused AdamW arithmetic
            module.optimizer_step(optimizer, max(float(tcfg["grad_clip_norm"]), 0.0))
        else:
            pred = (torch.sigmoid(score) > tcfg["pred_threshold"]).float()
            counts += torch.stack([(pred * labels).sum(), (pred * (1 - labels)).sum(), ((1 - pred) * labels).sum(),
                                   ((1 - pred) * (1 - labels)).sum()])
        sums += loss3.detach()
        n += 1
    stats = torch.cat([sums, counts, torch.tensor([float(n)], device=device)])
    if world > 1:
        torch.distributed.all_reduce(stats)
    stats = stats.cpu().tolist()  # the loop's only host sync
    n_tot = max(stats[7], 1.0)
    return {"total": stats[0] / n_tot, "bce": stats[1] / n_tot, "alignment": stats[2] / n_tot, "tp": stats[3], "fp": stats[4],
            "fn": stats[5], "tn": stats[6], "samples_per_s": n_smp / max(time.time() - t_loop, 1e-9)}


def run_train_ragged(module, text, batches, tcfg, device, optimizer, world=1):
    """One epoch of optimizer steps over RAGGED batches: up to ``train.samples_per_step`` (video, expression) samples of
    different (N, T, L) per step (sola_forward_train_ragged / sola_backward_ragged).  The reference steps once per sample
    (configs/mevis/default.yaml:37 batch_size 1, train.py:62-137); here a step minimises the MEAN over the batch of the
    per-sample totals of train.py:98-113 (each a mean over the sample's own tracks), i.e. its gradient is the average of the
    gradients the reference's next K batch-1 steps would start from, followed by ONE clipped AdamW update instead of K.
    With world > 1 the ranks' averages are averaged again by the all-reduce (effective batch K x world)."""
    pw, temp, aw = tcfg["positive_weight"], tcfg["temperature"], tcfg["alignment_weight"]
    module.train(True)
    sums = torch.zeros(3, device=device)
    n = n_smp = 0
    t_loop = time.time()
    for batch in DevicePrefetcher(batches, device):  # the next batch's tokens are uploaded while this one trains
        texts, pos = text.encode_ragged([s["expression"] for s in batch["samples"]])
        labels = torch.cat([(s["labels"][tcfg["positive_metric"]] > tcfg["positive_threshold"]).float() for s in batch["samples"]]).to(device)
        module.forward_ragged(batch["videos"], texts, batch["sample_video"])  # train mode + grad enabled: the differentiable path
        flat, tok, offs, counts = module.last_ragged
        loss = track_selection_losses_ragged(flat, tok, labels, pos, module.negative_token.weight, offs, counts, pw, temp, aw)
        optimizer.zero_grad(set_to_none=True)
        loss[:, 0].mean().backward()
        sdist.allreduce_gradient_arena(module, world)
        module.optimizer_step(optimizer, max(float(tcfg["grad_clip_norm"]), 0.0))  # clip + AdamW, one launch (bit-identical to clip_grad_norm_ + step)
        sums += loss.detach().sum(0)
        n += len(counts)
        n_smp += len(counts)
    stats = torch.cat([sums, torch.zeros(4, device=device), torch.tensor([float(n)], device=device)])
    if world > 1:
        torch.distributed.all_reduce(stats)
    stats = stats.cpu().tolist()  # the loop's only host sync
    n_tot = max(stats[7], 1.0)
    return {"total": stats[0] / n_tot, "bce": stats[1] / n_tot, "alignment": stats[2] / n_tot, "tp": 0.0, "fp": 0.0, "fn": 0.0, "tn": 0.0,
            "samples_per_s": n_smp / max(time.time() - t_loop, 1e-9)}


@torch.no_grad()
def run_split_ragged(module, text, batches, tcfg, device, world=1):
    """Validation / evaluation over ragged batches (sola_forward_ragged + sola_loss_ragged): the same per-sample numbers as
    ``run_split(train=False)`` - each sample's losses are means over its own tracks, as at the reference's batch size of 1
    (train.py:147-216, evaluator.py:88-112) - at up to ``ragged_max_samples`` samples per launch, the expressions of one
    video sharing the text-independent half of the network."""
    pw, temp, aw = tcfg["positive_weight"], tcfg["temperature"], tcfg["alignment_weight"]
    module.eval()
    sums = torch.zeros(3, device=device)
    counts4 = torch.zeros(4, device=device)  # TP FP FN TN
    bce_eval = torch.zeros(1, device=device)
    n = 0
    for batch in DevicePrefetcher(batches, device):  # the next batch's tokens are uploaded while this one is scored
        videos = batch["videos"]
        texts, pos = text.encode_ragged([s["expression"] for s in batch["samples"]])
        module.forward_ragged(videos, texts, batch["sample_video"])
        flat, tok, offs, counts = module.last_ragged
        labels = torch.cat([(s["labels"][tcfg["positive_metric"]] > tcfg["positive_threshold"]).float() for s in batch["samples"]]).to(device)
        loss = track_selection_losses_ragged(flat, tok, labels, pos, module.negative_token.weight, offs, counts, pw, temp, aw)
        sums += loss.sum(0)
        prob = torch.sigmoid(flat)
        pred = (prob > tcfg["pred_threshold"]).float()
        counts4 += torch.stack([(pred * labels).sum(), (pred * (1 - labels)).sum(), ((1 - pred) * labels).sum(),
                                ((1 - pred) * (1 - labels)).sum()])
        # the reference's evaluator feeds the SIGMOID-ed scores to binary_cross_entropy_with_logits (evaluator.py:101,107-111;
        # SURVEY appendix A): reported next to the train.py:98-104 value so eval JSONs of the two code bases can be compared
        w = torch.where(labels > 0, torch.full_like(labels, pw), torch.ones_like(labels))
        per_track = torch.nn.functional.binary_cross_entropy_with_logits(prob, labels, weight=w, reduction="none")
        bce_eval += torch.stack([t.mean() for t in torch.split(per_track, counts)]).sum()
        n += len(counts)
    stats = torch.cat([sums, counts4, torch.tensor([float(n)], device=device), bce_eval])
    if world > 1:
        torch.distributed.all_reduce(stats)
    stats = stats.cpu().tolist()
    n_tot = max(stats[7], 1.0)
    return {"total": stats[0] / n_tot, "bce": stats[1] / n_tot, "alignment": stats[2] / n_tot, "tp": stats[3], "fp": stats[4],
            "fn": stats[5], "tn": stats[6], "bce_evaluator_convention": stats[8] / n_tot, "samples": stats[7]}


def train(cfg):
    rank, local_rank, world = sdist.init_from_env()
    device = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    module = LanguageAlignedTrackSelectionModule(cfg["model"]).to(device)
    # Arithmetic of the training step: exact f32 like the reference unless asked otherwise (--precision f16x3 | f16 | bf16, or
    # train.precision in the YAML): the reduced-precision steps have no range guard, so they are opt-in
    tp = cfg.get("precision", cfg["train"].get("precision"))
    if tp is not None:
        module.train_precision = str(tp)
    if world > 1:  # identical initial weights on every rank
        for t in module.state_dict().values():
            torch.distributed.broadcast(t, src=0)
    text = TextEncoder(cfg["model"]["roberta_version"], cfg["model"]["lang_token_dim"], device,
                       allow_standin=bool(cfg.get("synthetic", False)))
    synthetic = cfg.get("synthetic", None)
    train_loader, _ = make_loader(cfg["dataset"], "train", rank, world, synthetic, cfg["model"])
    valid_batches, _ = make_ragged_batches(cfg["dataset"], "valid", rank, world, synthetic, cfg["model"])
    tcfg = cfg["train"]
    # fused=True: one multi-tensor kernel per step instead of torch's foreach passes (same update rule; 0.8 -> 0.3 ms per step)
    optimizer = torch.optim.AdamW(module.parameters(), lr=tcfg["lr"], fused=torch.cuda.is_available())
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=tcfg["lr_factor"], patience=tcfg["lr_patience"])
    n_epochs = int(cfg.get("n_epochs_override", tcfg["n_epochs"]))
    # --samples_per_step K (default: dataset.train.batch_size = 1, the reference's): K > 1 trains on ragged batches of K
    # variable-shape samples per optimizer step (run_train_ragged); K = 1 is the reference's loop, one sample per step
    per_step = int(cfg.get("samples_per_step", tcfg.get("samples_per_step", cfg["dataset"]["train"].get("batch_size", 1))))
    batches = None
    for epoch in range(n_epochs):
        t0 = time.time()
        if per_step > 1:
            batches, _ = make_ragged_train_batches(cfg["dataset"], rank, world, synthetic, cfg["model"], per_step, epoch, reuse=batches)
            tr = run_train_ragged(module, text, batches, tcfg, device, optimizer, world)
        else:
            tr = run_split(module, text, train_loader, tcfg, device, True, optimizer, world)
        if not math.isfinite(tr["total"]):  # the loops have no per-step host sync; an overflow shows in the epoch's sums
            raise SolaError(f"epoch {epoch + 1}: non-finite training loss with training precision {module.train_precision!r}"
                            + ("" if module.train_precision == "f32" else " - a value left the 16-bit operand range; rerun with --precision f32"))
        va = run_split_ragged(module, text, valid_batches, tcfg, device, world)
        scheduler.step(va["total"])
        if rank == 0:
            prec = va["tp"] / max(va["tp"] + va["fp"], 1.0)
            rec = va["tp"] / max(va["tp"] + va["fn"], 1.0)
            line = (f"EPOCH {epoch + 1} | train total {tr['total']:.4f} bce {tr['bce']:.4f} align {tr['alignment']:.4f} | "
                    f"valid total {va['total']:.4f} bce {va['bce']:.4f} align {va['alignment']:.4f} | precision {prec:.4f} "
                    f"recall {rec:.4f} | {time.time() - t0:.1f} s | train {tr['samples_per_s']:.0f} samples/s/rank incl. data + text"
                    + (f" ({per_step} samples per step)" if per_step > 1 else ""))
            print(line, flush=True)
            with open(os.path.join(cfg["results"]["output_dir"], "log.txt"), "a") as f:
                f.write(line + "\n")
            torch.save(module.state_dict(), os.path.join(cfg["results"]["output_dir"], f"epoch_{epoch + 1}.pth"))
    if world > 1:
        # every rank stepped on the same averaged gradient: the weights must be identical everywhere
        import hashlib

        h = hashlib.sha256()
        for t in module.state_dict().values():
            h.update(t.detach().cpu().numpy().tobytes())
        print(f"[rank {rank}] weights sha256 {h.hexdigest()[:16]}", flush=True)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    configs = load_configs("train")
    set_seed(42)
    train(configs)
