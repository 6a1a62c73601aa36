"""Config loading with the reference's CLI contract (train.py:254-292, eval.py:44-93, inference.py:100-147):
``--config <dir/name>`` selects ``configs/<dir/name>.yaml``; ``--eval_weight_epoch`` / ``--eval_pred_threshold`` exist for
eval/inference; every other ``--key value`` pair becomes a flat top-level ``configs[key]`` (ints, floats and booleans
coerced, bare flags -> True).  Output directories follow the reference layout:

    train      results.output_dir/<exp_name>/<train.data_name>/epoch_K.pth
    eval       results.eval_output_dir/<exp_name>/<valid.data_name>/pred_threshold_<t>/epoch_K/
    inference  results.test_output_dir/<exp_name>/<test.data_name>/pred_threshold_<t>/epoch_K/<video>/<exp>/<frame>.png
"""
from __future__ import annotations

import argparse
import os

import yaml


def _coerce(value: str):
    if value.replace(".", "", 1).isdigit():
        return float(value) if "." in value else int(value)
    if value.lower() in ("true", "false"):
        return value.lower() == "true"
    return value


def parse_overrides(unknown):
    out, i = {}, 0
    while i < len(unknown):
        tok = unknown[i]
        if tok.startswith("--"):
            if i + 1 < len(unknown) and not unknown[i + 1].startswith("--"):
                out[tok[2:]] = _coerce(unknown[i + 1])
                i += 2
            else:
                out[tok[2:]] = True
                i += 1
        else:
            i += 1
    return out


def load_configs(mode: str, argv=None, config_root="configs"):
    """mode in {"train", "eval", "inference"}."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default=None)
    if mode != "train":
        ap.add_argument("--eval_weight_epoch", type=int, default=None)
        ap.add_argument("--eval_pred_threshold", type=float, default=0.5)
    args, unknown = ap.parse_known_args(argv)
    assert args.config is not None, "config file must be provided"
    with open(os.path.join(config_root, f"{args.config}.yaml"), "r") as f:
        cfg = yaml.safe_load(f)
    cfg.update(parse_overrides(unknown))
    for k in list(cfg):  # build-side knobs: the synthetic stand-in dataset (--synthetic_samples / _tracks / _frames /
        if k.startswith("synthetic_") or k.startswith("ragged_"):  # _per_video / _ragged) and the ragged batcher's bounds
            cfg["dataset"][k] = cfg[k]
    res = cfg["results"]
    weight_dir = os.path.join(res["output_dir"], cfg["exp_name"], cfg["dataset"]["train"]["data_name"])
    if mode == "train":
        res["output_dir"] = weight_dir
        os.makedirs(weight_dir, exist_ok=True)
    else:
        assert args.eval_weight_epoch is not None, "--eval_weight_epoch must be provided"
        cfg["eval"]["weight_epoch"] = args.eval_weight_epoch
        cfg["eval"]["pred_threshold"] = args.eval_pred_threshold
        cfg["eval"]["weight_path"] = os.path.join(weight_dir, f"epoch_{args.eval_weight_epoch}.pth")
        split = "valid" if mode == "eval" else "test"
        key = "eval_output_dir" if mode == "eval" else "test_output_dir"
        thr = str(args.eval_pred_threshold).replace(".", "")  # 0.5 -> "05", as inference.py:140
        res[key] = os.path.join(res[key], cfg["exp_name"], cfg["dataset"][split]["data_name"], f"pred_threshold_{thr}",
                                f"epoch_{args.eval_weight_epoch}")
        os.makedirs(res[key], exist_ok=True)
    return cfg
