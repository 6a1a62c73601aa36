import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sola_amd import ops
B, N, Tp, D, H = 1, 64, 4, 1024, 8
rng = np.random.default_rng(0)
q, k, v = (torch.from_numpy(rng.standard_normal((B * N * Tp, D)).astype(np.float32)).cuda() for _ in range(3))
qs, ks, vs = (ops.cast_sp16(t) for t in (q, k, v))
a = ops.attention_split(qs, ks, vs, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp))
b = ops.decode_sp16(ops.attention_split(qs, ks, vs, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp), out_split=True))
e = (a - b).abs()
print("max err", float(e.max()), "by column mod 16:", [float(e[:, i::16].max()) for i in range(16)])
raw = ops.attention_split(qs, ks, vs, B * Tp, H, N, N, Tp, (N * Tp, 1, Tp), (N * Tp, 1, Tp), out_split=True)
h = raw.view(torch.float16).reshape(-1, D // 8, 16)
print("row0 block0 hi", h[0, 0, :8].tolist(), "lo", h[0, 0, 8:].tolist(), "f32", a[0, :8].tolist())
idx = torch.nonzero(e > 1e-5)
print("n bad", idx.shape[0], "of", e.numel())
hh = raw.view(torch.float16).reshape(a.shape[0], D // 8, 2, 8)
for r, c in idx[:6].tolist():
    print("row", r, "col", c, "f32", float(a[r, c]), "hi", float(hh[r, c // 8, 0, c % 8]), "lo", float(hh[r, c // 8, 1, c % 8]), "g4", (c % 16) // 4, "j", c % 4,
          "head", c // 128)
rows = sorted(set(i[0] for i in idx.tolist()))
print("bad rows", rows[:20], "...")
