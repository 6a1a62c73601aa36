#!/usr/bin/env python3
"""Whole-batch parity of the benched batches: every row of a 256-sample batch at the headline shape (T=32, N=64, L=16; seeds
1000 = bench.py's rank-0 batch, 1001, 1002) and of a 32-sample batch at C4 (T=128, N=128) against the fp32 PyTorch-CPU oracle, in
the exact-f32 and the split-f16 (f16x3) modes.  Prints one JSON line per (shape, seed): the largest |logit| / |token| error over
ALL rows per mode and whether every thresholded selection agrees.

    python tools/batch_error.py [seeds, comma separated = 1000,1001,1002] [c4 = 1] [float64 check of the first seed = 1]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sola_oracle  # noqa: E402  (checker only)
from sola_amd import synth  # noqa: E402
from sola_amd.module import LanguageAlignedTrackSelectionModule  # noqa: E402

seeds = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000,1001,1002").split(",")]
with_c4 = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
with_f64 = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
cfg = synth.DEFAULT_MODEL_CFG
sd = synth.make_state_dict(cfg, 42)
tsd = sola_oracle.to_torch_state(sd)
mods = {}
for prec in ("f32", "f16x3"):
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m = m.cuda().eval()
    m.precision = prec
    mods[prec] = m
torch.set_num_threads(min(32, os.cpu_count() or 1))
cases = [("NS", 256, 64, 32, 16, s) for s in seeds] + ([("C4", 32, 128, 128, 16, 2000)] if with_c4 else [])
for tag, B, N, T, L, seed in cases:
    inp = synth.make_inputs(cfg, B, N, T, L, seed=seed)
    t0 = time.perf_counter()
    ref_sm, ref_st = [], []
    step = 16 if tag == "NS" else 2
    for b in range(0, B, step):
        rsm, rst = sola_oracle.forward(tsd, cfg, inp["object_tokens"][b:b + step], inp["lang_tokens"][b:b + step])
        ref_sm.append(rsm.numpy()); ref_st.append(rst.numpy())
    ref_sm, ref_st = np.concatenate(ref_sm), np.concatenate(ref_st)
    row = {"shape": tag, "B": B, "N": N, "T": T, "seed": seed, "oracle_s": round(time.perf_counter() - t0, 1)}
    ref64 = None
    if with_f64 and tag == "NS" and seed == seeds[0]:  # the fp32 oracle (= the reference's arithmetic) has its own rounding noise: float64 says whose
        t0 = time.perf_counter()
        tsd64 = sola_oracle.to_torch_state(sd, torch.float64)
        r64 = [sola_oracle.forward(tsd64, cfg, inp["object_tokens"][b:b + step], inp["lang_tokens"][b:b + step], dtype=torch.float64)[0].numpy()
               for b in range(0, B, step)]
        ref64 = np.concatenate(r64)
        e = np.abs(ref_sm - ref64).max(axis=1)
        row["fp32_oracle_vs_float64"] = {"max_abs_logit_err": float(e.max()), "rows_above_5e-4": int((e > 5e-4).sum()), "mean_row_max_err": float(e.mean()),
                                         "float64_s": round(time.perf_counter() - t0, 1)}
    for prec, m in mods.items():
        with torch.no_grad():
            sm, st = m(torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda())
        sm, st = sm.cpu().numpy(), st.cpu().numpy()
        e_rows = np.abs(sm - ref_sm).max(axis=1)
        row[prec] = {"max_abs_logit_err_vs_oracle": float(e_rows.max()), "rows_above_5e-4": int((e_rows > 5e-4).sum()),
                     "mean_row_max_err": float(e_rows.mean()), "max_abs_token_err_vs_oracle": float(np.abs(st - ref_st).max()),
                     "selections_equal": bool(np.array_equal(sm > 0, ref_sm > 0)), "guard_bits": m.split_fallbacks()[1]}
        if ref64 is not None:
            e64 = np.abs(sm - ref64).max(axis=1)
            row[prec]["vs_float64"] = {"max_abs_logit_err": float(e64.max()), "rows_above_5e-4": int((e64 > 5e-4).sum()), "mean_row_max_err": float(e64.mean())}
    print(json.dumps(row), flush=True)
