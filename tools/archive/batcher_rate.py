"""RaggedBatcher alone on the synthetic ragged mix: samples/s by reader threads and with / without page-locking the tokens.
usage: batcher_rate.py [samples = 512]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd.data import SyntheticTracks, RaggedBatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.cuda.init()
ds = SyntheticTracks(n_samples=n, token_dim=256, seed=0, with_labels=True, per_video=4, ragged=True)
order = torch.randperm(n, generator=torch.Generator().manual_seed(1)).tolist()
for nw in (8, 16, 32, 64):
    for pin in (False, True):
        b = RaggedBatcher(ds, order, 64, max_rows=1 << 62, num_workers=nw, pin=pin)
        sum(1 for _ in b)
        def drain():
            mb = 0.0
            for x in b:
                mb += sum(v.numel() * 4 for v in x["videos"]) / 1e6
                if "pinned_pool" in x: x["pinned_pool"].give_back(x["pinned_bufs"])  # what DevicePrefetcher does once the upload is done
            return mb
        drain()
        t0 = time.perf_counter(); mb = drain(); dt = time.perf_counter() - t0
        print(f"reader threads {nw:3d} pin {int(pin)}: {n / dt:7.0f} samples/s  {mb / dt / 1e3:5.2f} GB/s of tokens" +
              (f"  pool hits {b.pool.hits} misses {b.pool.misses}" if b.pool else ""), flush=True)
