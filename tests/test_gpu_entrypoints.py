"""End-to-end run of the three entry points on the GPU with the synthetic stand-in dataset: train one epoch (forward,
losses, HIP backward, multi-tensor clip, AdamW), save epoch_1.pth in the reference layout, then eval and inference load it."""
import json
import os
import subprocess
import sys

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_eval_inference_roundtrip(tmp_path):
    os.makedirs(tmp_path / "configs" / "mevis")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "mevis", "default.yaml")))
    cfg["dataset"]["track_root"] = str(tmp_path / "no_such_dir")
    yaml.safe_dump(cfg, open(tmp_path / "configs" / "mevis" / "default.yaml", "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SOLA_PRECISION", None)  # the entry points' own default (f16x3)
    common = ["--config", "mevis/default", "--synthetic", "true", "--synthetic_samples", "6", "--synthetic_tracks", "8", "--synthetic_frames", "16"]

    def run(script, *extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), *common, *extra], cwd=tmp_path, env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return r.stdout

    out = run("train.py", "--n_epochs_override", "2")
    wdir = tmp_path / "SOLA" / "TRAIN" / "default" / "mevis"
    assert (wdir / "epoch_1.pth").exists() and (wdir / "epoch_2.pth").exists() and "EPOCH 2" in out
    sd = torch.load(wdir / "epoch_2.pth", map_location="cpu", weights_only=True)
    assert len(sd) == 84 and all(torch.isfinite(v).all() for v in sd.values())
    sd1 = torch.load(wdir / "epoch_1.pth", map_location="cpu", weights_only=True)
    assert any(not torch.equal(sd[k], sd1[k]) for k in sd)  # the optimizer moved the weights
    run("eval.py", "--eval_weight_epoch", "2")
    m = json.load(open(tmp_path / "SOLA" / "EVAL" / "default" / "mevis" / "pred_threshold_05" / "epoch_2" / "track_metrics.json"))
    assert m["tp"] + m["fp"] + m["fn"] + m["tn"] == 6 * 8 and m["total"] > 0
    assert m["samples"] == 6 and m["text_encoder"] == "hashed-standin" and m["bce_evaluator_convention"] > 0
    run("inference.py", "--eval_weight_epoch", "2")
    inf = tmp_path / "SOLA" / "INFERENCE" / "default" / "mevis" / "pred_threshold_05" / "epoch_2"
    assert len(list(inf.rglob("*_pred.npy"))) == 6
    # a MeViS-like mix of shapes (N in [8,80], T in [20,200], four expressions per video): 24 samples in ONE ragged launch
    # give the decisions of 24 one-sample launches (the reference's batch size of 1)
    import numpy as np
    import shutil

    rag = ["--config", "mevis/default", "--synthetic", "true", "--synthetic_samples", "24", "--synthetic_ragged", "true"]
    preds = {}
    for tag, extra in (("one", ["--ragged_max_samples", "1"]), ("many", ["--ragged_max_samples", "64"])):
        shutil.rmtree(inf)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), *rag, "--eval_weight_epoch", "2", *extra], cwd=tmp_path,
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        assert ("24 samples / 6 video passes in 1 ragged calls" in r.stdout) == (tag == "many"), r.stdout[-500:]
        preds[tag] = {str(f.relative_to(inf)): np.load(f) for f in sorted(inf.rglob("*_pred.npy"))}
    assert len(preds["one"]) == 24 and preds["one"].keys() == preds["many"].keys()
    assert len({v.shape for v in preds["many"].values()}) > 1  # really ragged
    for k in preds["one"]:
        np.testing.assert_array_equal(preds["one"][k], preds["many"][k], err_msg=k)


def test_train_on_ragged_batches(tmp_path):
    """train.py --samples_per_step 8 on a mix of shapes (N in [8,80], T in [20,200]): three ragged optimizer steps per epoch
    over 20 samples (8 + 8 + 4), the validation pass, a finite checkpoint that moved."""
    os.makedirs(tmp_path / "configs" / "mevis")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "mevis", "default.yaml")))
    cfg["dataset"]["track_root"] = str(tmp_path / "no_such_dir")
    yaml.safe_dump(cfg, open(tmp_path / "configs" / "mevis" / "default.yaml", "w"))
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SOLA_PRECISION", None)  # the entry points' own default (f16x3)
    args = ["--config", "mevis/default", "--synthetic", "true", "--synthetic_samples", "20", "--synthetic_ragged", "true",
            "--samples_per_step", "8", "--n_epochs_override", "2"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "EPOCH 2" in r.stdout and "8 samples per step" in r.stdout, r.stdout[-800:]
    wdir = tmp_path / "SOLA" / "TRAIN" / "default" / "mevis"
    sd1 = torch.load(wdir / "epoch_1.pth", map_location="cpu", weights_only=True)
    sd2 = torch.load(wdir / "epoch_2.pth", map_location="cpu", weights_only=True)
    assert all(torch.isfinite(v).all() for v in sd2.values())
    assert any(not torch.equal(sd1[k], sd2[k]) for k in sd1)


def test_bench_two_ranks_code_path(tmp_path):
    """bench.py under torch.distributed.run with 2 ranks (gloo, both on cuda:0 - RCCL refuses two ranks per device): the
    barrier / max-over-ranks / whole-job aggregation path the driver uses at N > 1 prints one well-formed JSON line."""
    env = dict(os.environ, PYTHONPATH=ROOT, SOLA_BENCH_BACKEND="gloo")
    env.pop("SOLA_PRECISION", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--cpu-seconds", "0", "--train-steps", "2"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["unit"] == "samples/s"
    assert out["config"]["sharding"] == "per-sample x2" and "roofline" in out and "cpu_baseline" not in out
    # the driver's contract at N > 1: every key of the line, the backend's own view of the job, the training leg
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True and out["vs_baseline"] is None and out["data"] == "synthetic"
    assert out["metric"].startswith("track-selection forward+loss samples/sec at (T=32,N=64,d=256)")
    assert out["config"]["world_size_reported_by_backend"] == 2 and out["config"]["collective_backend"] == "gloo"
    assert out["config"]["batch_per_gpu"] == 4 and abs(out["value"] - 2 * 4 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]  # whole-job samples/s
    if out["roofline"] is not None:
        assert out["roofline"]["frac"] > 0 and out["roofline"]["bound"] == "mfma"
    assert out["roofline_attention"]["bound"] == "hbm" and out["roofline_attention"]["frac"] > 0
    # the multi-rank training leg: the one collective of the path, with and without overlap, and the per-rank ragged inference leg
    td = out["training_step_dist"]
    assert td["world_size"] == 2 and td["backend"] == "gloo"
    for k in ("overlap", "no_overlap", "no_collective"):
        assert td[k]["value"] > 0 and td[k]["ms_per_step"] > 0
    assert td["allreduce_alone"]["bytes"] >= 4 * 32_980_000 and td["allreduce_alone"]["busbw_GBps"] > 0
    assert td["ragged_inference"]["value"] > 0


def test_staged_batches_equal_unstaged_ones_and_buffers_are_recycled():
    """RaggedBatcher(pin=True): reader threads copy each sample's tokens into PinnedPool buffers (page-locked, recycled once the
    uploader's copy event has completed).  The uploaded videos must equal the unstaged ones bit for bit over two passes of a ragged
    synthetic set - a buffer handed back too early would be overwritten by a later sample before its upload - and the second pass
    must be served from the pool."""
    sys.path.insert(0, ROOT)
    from sola_amd.data import DevicePrefetcher, RaggedBatcher, SyntheticTracks

    ds = SyntheticTracks(n_samples=96, token_dim=256, seed=3, with_labels=True, per_video=2, ragged=True)
    order = list(range(96))
    plain = [[v.clone() for v in b["videos"]] for b in RaggedBatcher(ds, order, 16, max_rows=1 << 62)]
    staged = RaggedBatcher(ds, order, 16, max_rows=1 << 62, num_workers=8, pin=True)
    dev = torch.device("cuda", 0)
    for _ in range(2):
        n = 0
        for b, ref in zip(DevicePrefetcher(staged, dev), plain):
            assert "pinned_bufs" not in b and len(b["videos"]) == len(ref)
            for v, r in zip(b["videos"], ref):
                assert v.is_cuda and torch.equal(v.cpu(), r)
            n += 1
        assert n == len(plain)
    assert staged.pool.hits > 0 and staged.pool.misses <= 96
