#!/usr/bin/env python3
"""The bench's ragged legs alone (MeViS-like mix, 128 samples per launch), with the per-kernel split."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sola_amd import synth
from sola_amd.module import LanguageAlignedTrackSelectionModule
cfg = synth.DEFAULT_MODEL_CFG
m = LanguageAlignedTrackSelectionModule(cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, 42).items()})
m = m.cuda().eval(); m.precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"; m.ws_policy = "always"
from sola_amd import _lib
# attention routing of the ragged batches: LDS shapes only / + resident-K/V shape (attn_res 2) / + register-only shape (attn_reg 2),
# then the split-f16 attention threshold (units of more keys than this take q/k/v as split pairs): 64 = rounds 1-2, 128 = default
for tag, res, reg, smk in (("lds", 1, 1, 128), ("res", 2, 1, 128), ("res+reg", 2, 2, 128), ("lds split>64", 1, 1, 64), ("lds", 1, 1, 128), ("lds split>64", 1, 1, 64)):
    _lib.check(_lib.lib().sola_tune(b"attn_res", res), "tune"); _lib.check(_lib.lib().sola_tune(b"attn_reg", reg), "tune")
    _lib.check(_lib.lib().sola_tune(b"attn_split_min_keys", smk), "tune")
    r = bench.ragged_leg(cfg, m, torch.device("cuda", 0), 6, 288.0)
    for k in ("one_expression_per_video", "four_expressions_per_video"):
        print(tag, k, json.dumps({kk: r[k][kk] for kk in ("value", "ms_per_launch")}), "attn", r[k]["kernel_ms_per_launch"]["attn"], "gn", r[k]["kernel_ms_per_launch"]["group_norm"],
              "gemm", r[k]["kernel_ms_per_launch"].get("gemm_split256"), flush=True)
_lib.lib().sola_tune(b"attn_res", 1); _lib.lib().sola_tune(b"attn_reg", 1); _lib.lib().sola_tune(b"attn_split_min_keys", 128)
