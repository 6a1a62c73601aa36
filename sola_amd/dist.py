"""Multi-GPU plumbing for the track-selection path: one process per GPU, RCCL over xGMI via torch.distributed.

The path shards by independent (video, expression) samples: forward, loss and selection need no
collective at all.  Training adds exactly one exchange per optimizer step - the all-reduce of the 32.98M-element
gradient - issued as a few large flat buckets (xGMI is point-to-point, 7 links per GPU: large messages, no per-tensor
calls).  The reference itself has no distributed code; its only sharding idiom is the ``idx % n_pid == pid`` stride of
track_generation/generate_tokens_gdino.py:97, which ``shard_indices`` keeps.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:  # SOLA_DIST_BACKEND=gloo: several ranks on one GPU (tests; RCCL refuses two ranks per device)
            backend = os.environ.get("SOLA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def shard_indices(n_samples, rank, world, pad=False):
    """Static round-robin ownership: sample i belongs to rank i % world.

    ``pad=True`` (training): every rank gets exactly ceil(n / world) indices, the short shards being completed by
    wrapping around to the first samples (torch's DistributedSampler rule with drop_last=False).  Training issues one
    gradient all-reduce per optimizer step, so ranks with different step counts would pair a 131.9 MB gradient bucket
    with another rank's 8-float statistics reduce - a hang or silent corruption on RCCL."""
    if pad and world > 1 and n_samples > 0:
        total = -(-n_samples // world) * world
        order = [i % n_samples for i in range(total)]  # the index list, completed from its own head
        return order[rank::world]
    return list(range(rank, n_samples, world))


def allreduce_gradients(params, world=None, bucket_bytes=64 << 20, average=True):
    """Sum (and average) ``p.grad`` over all ranks with a handful of flat-bucket all-reduces.

    Buckets are filled in parameter order, so every rank issues identical collectives.  Returns the number of
    collectives issued.  With world == 1 this is a no-op."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    n_coll = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, n_coll
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        n_coll += 1
        bucket, size = [], 0

    for g in grads:
        nb = g.numel() * g.element_size()
        if size + nb > bucket_bytes and bucket:
            flush()
        bucket.append(g)
        size += nb
    flush()
    return n_coll


def allreduce_gradient_arena(module, world=None, average=True, overlap=True):
    """The training step's one collective, in place and overlapped: the module keeps all gradients in ONE flat arena laid
    out in the order sola_backward finishes them (layer n-1 ... layer 0, encoder: n_layers + 1 buckets of 30-50 MB at the
    default size; negative_token.weight, whose gradient autograd completes later on the caller's stream, sits behind them - xGMI is point-to-point, so few large messages).  Called right after ``loss.backward()``
    has ENQUEUED the backward, it makes a side stream wait (device-side) for each bucket's completion event and issues that
    bucket's all-reduce there, so the layer buckets travel while the encoder's backward still runs; the caller's stream
    then waits for the side stream.  Every rank issues the same collectives in the same order.  Returns their number.

    Falls back to ``allreduce_gradients`` when the last backward did not write into the arena (gradient accumulation)."""
    from ._lib import check, lib

    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return 0
    if not getattr(module, "_grads_in_arena", False):
        return allreduce_gradients(module.parameters(), world, average=average)
    buckets = module.grad_buckets()
    dev = buckets[0].device
    nccl = dist.get_backend() == "nccl"
    main = torch.cuda.current_stream(dev)
    side = getattr(module, "_comm_stream", None)
    if side is None or side.device != dev:
        side = module._comm_stream = torch.cuda.Stream(device=dev)
    if not overlap:
        side = main
    for k, flat in enumerate(buckets):
        if overlap:
            check(lib().sola_backward_wait_bucket(module._ctx, k, side.cuda_stream), "sola_backward_wait_bucket")
        with torch.cuda.stream(side):
            if nccl and average:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG)  # RCCL averages in the collective
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                if average:
                    flat.div_(world)
    if overlap:
        main.wait_stream(side)
    # Parameters with a gradient path outside the network - negative_token.weight feeds the alignment loss directly
    # (train.py:92) - live behind the last bucket, outside every bucket's range (module._grad_layout): autograd adds the loss's
    # gradient to that slot on the caller's stream after the backward, possibly into a tensor of its own.  Whatever tensor
    # holds the parameter's gradient now is reduced HERE, on the caller's stream, behind that add (128 KB).
    tail = set(getattr(module, "_grad_tail", []))
    named = dict(module.named_parameters())
    extra = [named[k] for k in tail if named[k].grad is not None]
    # ... and anything else autograd moved out of its arena slot (a parameter used twice)
    extra += [p for key, p in named.items() if key not in tail and p.grad is not None
              and p.grad.data_ptr() != module._grad_view(key).data_ptr()]
    n_extra = allreduce_gradients(extra, world, average=average) if extra else 0
    module._last_grad_sq = None  # (the in-place collective also bumped the arena's version counter, which every view shares)
    return len(buckets) + n_extra


def gather_scores(local, rank, world):
    """Collect per-rank python objects (e.g. {sample_idx: scores}) on every rank; no tensor data path involved."""
    if world == 1:
        return [local]
    out = [None] * world
    dist.all_gather_object(out, local)
    return out
