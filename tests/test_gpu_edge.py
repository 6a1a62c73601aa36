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


def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 71)), int(rng.integers(1, 21))) for _ in range(n)]


@pytest.mark.parametrize("shape", _random_shapes(12, 2024))
def test_random_shapes_both_precisions(model, shape):
    """Seeded random (B, N, T, L): every kernel-selection branch (packed / shared / wide attention, 64x64 / split-K /
    direct-to-LDS GEMMs, wave / block GroupNorm units) is reached by some of them."""
    test_edge_shapes_both_precisions(model, shape)
