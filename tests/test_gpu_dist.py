"""Multi-GPU training path exercised with two ranks on ONE GPU (gloo backend: RCCL refuses two ranks per device; the data
path - flat gradient arena, per-bucket completion events, in-place all-reduce on a side stream, identical clip + AdamW on
every rank - is the same code the 8-GPU run takes with backend nccl)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(cfg, sd):
    from sola_amd.module import LanguageAlignedTrackSelectionModule

    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.cuda().train()
    m.dropout_p = 0.0
    m.attention_dropout_p = 0.0
    return m


def _loss(m, inp):
    from sola_amd.loss import track_selection_losses

    c = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    sm, st = m(c["object_tokens"], c["lang_tokens"])
    neg = m.negative_token.weight.clone().unsqueeze(0).repeat(c["lang_tokens"].shape[0], 1, 1)
    return track_selection_losses(sm, st, c["labels"], c["pos_tokens"], neg, 1.5, 0.07, 0.3)[0]


def _worker(rank, world, port, q, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from sola_amd import dist as sdist
    from sola_amd import synth

    torch.cuda.set_device(0)
    sdist.init_from_env("gloo")
    cfg = synth.SMALL_MODEL_CFG
    m = _make(cfg, synth.make_state_dict(cfg, 42))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    for step in range(2):
        inp = synth.make_inputs(cfg, 2, 6, 16, 5, seed=10 * step + rank)  # every rank its own samples
        opt.zero_grad(set_to_none=True)
        _loss(m, inp).backward()
        # autograd adopted the arena views without a copy: every p.grad lives in the arena, except the negative tokens', which
        # autograd sums with the alignment loss's direct contribution into a tensor of its own
        outside = [k for k, p in m.named_parameters() if p.grad.data_ptr() != m._grad_view(k).data_ptr()]
        assert m._grads_in_arena and outside == ["negative_token.weight"], outside
        # ... and its arena slot lies outside every bucket's flat range: the overlapped in-place collectives on the side stream
        # never touch memory that autograd's add on the main stream reads (that was a race)
        o, cnt, _shape = m._grad_spans["negative_token.weight"]
        assert all(not (a < o + cnt and o < b) for a, b in m._grad_buckets), (o, cnt, m._grad_buckets)
        assert m._grad_buckets[-1][1] <= o
        n = sdist.allreduce_gradient_arena(m, world, overlap=overlap)
        assert n == cfg["n_layers"] + 2  # the arena's buckets + one small collective for the straggler
        gnd = m.get_grad_norm_dict()
        m.clip_grad_norm_(0.5 * gnd["total_grad_norm"])  # always clips: the clip factor must be identical on both ranks
        opt.step()
    torch.cuda.synchronize()
    q.put((rank, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, gnd["total_grad_norm"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_step_identically_and_match_the_averaged_gradient(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (_, w0, n0), (_, w1, n1) = res
    assert n0 == n1
    for k in w0:
        np.testing.assert_array_equal(w0[k], w1[k], err_msg=k)  # identical weights on both ranks after two steps
    # the same two steps in ONE process on the averaged gradient of the two ranks' samples
    from sola_amd import synth

    cfg = synth.SMALL_MODEL_CFG
    m = _make(cfg, synth.make_state_dict(cfg, 42))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    for step in range(2):
        grads = None
        for rank in range(2):
            opt.zero_grad(set_to_none=True)
            _loss(m, synth.make_inputs(cfg, 2, 6, 16, 5, seed=10 * step + rank)).backward()
            g = [p.grad.clone() for p in m.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        opt.zero_grad(set_to_none=True)
        for p, g in zip(m.parameters(), grads):
            p.grad = g / 2
        gnd = m.get_grad_norm_dict()
        m.clip_grad_norm_(0.5 * gnd["total_grad_norm"])
        opt.step()
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.detach().cpu().numpy(), w0[k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_gradient_accumulation_does_not_alias_the_arena():
    """Two backwards without zero_grad in between: the second must not overwrite the gradient autograd accumulates into."""
    from sola_amd import synth

    cfg = synth.SMALL_MODEL_CFG
    m = _make(cfg, synth.make_state_dict(cfg, 42))
    a, b = synth.make_inputs(cfg, 1, 5, 16, 4, seed=1), synth.make_inputs(cfg, 1, 5, 16, 4, seed=2)
    _loss(m, a).backward()
    ga = [p.grad.clone() for p in m.parameters()]
    _loss(m, b).backward()  # accumulates
    assert not m._grads_in_arena
    gab = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    _loss(m, b).backward()
    assert m._grads_in_arena
    for x, y, z in zip(ga, gab, [p.grad for p in m.parameters()]):
        torch.testing.assert_close(y, x + z, rtol=1e-5, atol=1e-7)


def test_train_py_two_ranks_odd_sample_count(tmp_path):
    """The real entry point under torchrun with 2 ranks and 7 training samples (uneven shards): both ranks
    run the same number of optimizer steps, finish, and end with identical weights."""
    os.makedirs(tmp_path / "configs" / "mevis")
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "mevis", "default.yaml")))
    cfg["dataset"]["track_root"] = str(tmp_path / "no_such_dir")
    yaml.safe_dump(cfg, open(tmp_path / "configs" / "mevis" / "default.yaml", "w"))
    env = dict(os.environ, PYTHONPATH=ROOT, SOLA_DIST_BACKEND="gloo")
    env.pop("SOLA_PRECISION", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), "--config", "mevis/default", "--synthetic", "true",
                        "--synthetic_samples", "7", "--synthetic_tracks", "8", "--synthetic_frames", "16", "--n_epochs_override", "1"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    shas = [ln.split()[-1] for ln in r.stdout.splitlines() if "weights sha256" in ln]
    assert len(shas) == 2 and shas[0] == shas[1], r.stdout[-1500:]
    assert "EPOCH 1" in r.stdout
    # the same with RAGGED optimizer steps: 9 variable-shape samples on 2 ranks, 4 samples per step -> 5 per rank (padded shard)
    # = two steps per rank, each ending in the gradient all-reduce
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), "--config", "mevis/default", "--synthetic", "true",
                        "--synthetic_samples", "9", "--synthetic_ragged", "true", "--samples_per_step", "4", "--n_epochs_override", "1"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    shas = [ln.split()[-1] for ln in r.stdout.splitlines() if "weights sha256" in ln]
    assert len(shas) == 2 and shas[0] == shas[1], r.stdout[-1500:]
    assert "4 samples per step" in r.stdout
