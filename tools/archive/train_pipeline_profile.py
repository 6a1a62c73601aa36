"""Where train.py --samples_per_step spends its wall time on the synthetic ragged mix: the batcher alone (reader threads that also pin),
+ pinning / upload (DevicePrefetcher), + text stand-in, + the training step itself.

    python tools/train_pipeline_profile.py [workers = 16] [samples = 512]
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import synth
from sola_amd.data import SyntheticTracks, RaggedBatcher, DevicePrefetcher
from sola_amd.loss import track_selection_losses_ragged
from sola_amd.module import LanguageAlignedTrackSelectionModule
from sola_amd.text import TextEncoder

def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    dev = torch.device("cuda", 0)
    cfg = synth.DEFAULT_MODEL_CFG
    ds = SyntheticTracks(n_samples=n, token_dim=256, seed=0, with_labels=True, per_video=4, ragged=True)
    order = torch.randperm(n, generator=torch.Generator().manual_seed(1)).tolist()
    _b = RaggedBatcher(ds, order, 64, max_rows=1 << 62, num_workers=nw, pin=True)  # ONE batcher: its worker processes persist across the passes
    mk = lambda: _b
    sum(1 for _ in mk())  # start the workers (spawned interpreters) outside the timed passes

    t0 = time.perf_counter(); nb = sum(1 for _ in mk()); t_batch = time.perf_counter() - t0
    mb = sum(v.numel() * 4 for b in mk() for v in b["videos"]) / 1e6
    t0 = time.perf_counter()
    for b in DevicePrefetcher(mk(), dev):
        pass
    torch.cuda.synchronize(); t_pref = time.perf_counter() - t0
    text = TextEncoder("none", cfg["lang_token_dim"], dev, allow_standin=True)
    t0 = time.perf_counter()
    for b in DevicePrefetcher(mk(), dev):
        text.encode_ragged([s["expression"] for s in b["samples"]])
    torch.cuda.synchronize(); t_text = time.perf_counter() - t0
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
    m = m.to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5, fused=True)
    def run():
        for b in DevicePrefetcher(mk(), dev):
            texts, pos = text.encode_ragged([s["expression"] for s in b["samples"]])
            objs = [b["videos"][v] for v in b["sample_video"]]
            labels = torch.cat([(s["labels"]["iou"] > 0.5).float() for s in b["samples"]]).to(dev)
            opt.zero_grad(set_to_none=True)
            m.forward_ragged(objs, texts)
            flat, tok, offs, counts = m.last_ragged
            loss = track_selection_losses_ragged(flat, tok, labels, pos, m.negative_token.weight, offs, counts, 1.5, 0.07, 0.3)
            loss[:, 0].mean().backward()
            m.clip_grad_norm_(1.0)
            opt.step()
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{n} samples in {nb} batches, {mb / 2:.0f} MB of tokens, {nw} workers: batcher alone {n / t_batch:.0f} samples/s; + pin / upload {n / t_pref:.0f}; "
          f"+ text stand-in {n / t_text:.0f}; + training step {n / t_all:.0f} samples/s")


if __name__ == "__main__":  # spawned DataLoader workers re-import this module
    main()
