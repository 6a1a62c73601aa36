"""Edge shapes of the whole path (single track / single frame / single text token, N beyond one wave tile, T' = 100 =
max_temporal_length, odd sizes) in both arithmetic modes, against the float32 oracle computed here."""
import numpy as np
import pytest
import torch

from oracle import sola_oracle
from sola_amd import SolaError, synth
from sola_amd.module import LanguageAlignedTrackSelectionModule

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1, 1), (1, 2, 3, 1), (3, 1, 9, 2), (1, 300, 8, 4), (1, 8, 800, 16), (2, 7, 33, 77), (1, 17, 5, 3), (5, 3, 2, 1)]


@pytest.fixture(scope="module")
def model():
    cfg = synth.DEFAULT_MODEL_CFG
    sd = synth.make_state_dict(cfg, 42)
    m = LanguageAlignedTrackSelectionModule(cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return m.cuda().eval(), sd, cfg


@pytest.mark.parametrize("shape", SHAPES)
def test_edge_shapes_both_precisions(model, shape):
    m, sd, cfg = model
    B, N, T, L = shape
    inp = synth.make_inputs(cfg, B, N, T, L, 3)
    rsm, rst = sola_oracle.forward(sd, cfg, inp["object_tokens"], inp["lang_tokens"])
    for prec in ("f32", "f16x3"):
        m.precision = prec
        with torch.no_grad():
            sm, st = m(torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda())
        assert tuple(sm.shape) == (B, N) and tuple(st.shape) == (B, N, cfg["lang_token_dim"])
        assert float((sm.cpu() - rsm).abs().max()) <= 1e-3 and float((st.cpu() - rst).abs().max()) <= 1e-3
        assert torch.equal(torch.sigmoid(sm.cpu()) > 0.5, torch.sigmoid(rsm) > 0.5)
    m.precision = "f32"


def test_bad_inputs_raise(model):
    m, _, cfg = model
    obj = torch.zeros(1, 4, 8, cfg["object_token_dim"], device="cuda")
    lang = torch.zeros(1, 3, cfg["lang_token_dim"], device="cuda")
    for grad in (False, True):  # the inference call and the autograd (training) call validate alike
        with torch.set_grad_enabled(grad):
            with pytest.raises(SolaError):
                m(obj.cpu(), lang)                      # CPU tensor: no fallback path
            with pytest.raises(SolaError):
                m(obj[..., :100], lang)                 # wrong token width
            with pytest.raises(SolaError):
                m(obj, torch.zeros(2, 3, cfg["lang_token_dim"], device="cuda"))  # batch mismatch
            with pytest.raises(SolaError):
                m(obj[:, :0], lang)                     # no tracks
            with pytest.raises(SolaError):
                m(obj[0], lang)                         # missing batch dimension


def test_bf16_is_a_training_mode(model):
    """Library precision 3 (bfloat16 GEMM operands) exists for the training step only: the inference entry points refuse it
    (SOLA_ERR_ARG), and a module set to "bf16" runs its inference calls on the default split-f16 kernels."""
    import ctypes as C
    from sola_amd import _lib
    m, sd, cfg = model
    inp = synth.make_inputs(cfg, 1, 6, 16, 5, 9)
    obj, lang = torch.from_numpy(inp["object_tokens"]).cuda(), torch.from_numpy(inp["lang_tokens"]).cuda()
    m.precision = "f16x3"
    with torch.no_grad():
        want = [t.clone() for t in m(obj, lang)]
    m.precision = "bf16"
    try:
        with torch.no_grad():
            got = m(obj, lang)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        _lib.check(_lib.lib().sola_set_precision(m._ctx, 3), "sola_set_precision")
        m._ctx_precision = None  # the module re-applies its own setting on the next call
        sm, st = torch.empty(1, 6, device="cuda"), torch.empty(1, 6, cfg["lang_token_dim"], device="cuda")
        ws = torch.empty(int(_lib.lib().sola_workspace_bytes(m._ctx, 1, 6, 16, 5)), dtype=torch.uint8, device="cuda")
        rc = _lib.lib().sola_forward(m._ctx, _lib.ptr(obj), _lib.ptr(lang), 1, 6, 16, 5, _lib.ptr(sm), _lib.ptr(st), _lib.ptr(ws),
                                     C.c_size_t(ws.numel()), None)
        assert rc != 0 and b"TRAINING mode" in _lib.lib().sola_last_error()
    finally:
        m.precision = "f32"


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 71)), int(rng.integers(1, 21))) for _ in range(n)]


@pytest.mark.parametrize("shape", _random_shapes(12, 2024))
def test_random_shapes_both_precisions(model, shape):
    """Seeded random (B, N, T, L): every kernel-selection branch (packed / shared / wide attention, 64x64 / split-K /
    direct-to-LDS GEMMs, wave / block GroupNorm units) is reached by some of them."""
    test_edge_shapes_both_precisions(model, shape)
