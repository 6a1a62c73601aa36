"""Training path (forward that saves activations + HIP backward).  Not available yet in this build:
the call fails loudly rather than falling back to PyTorch autograd."""
from ._lib import SolaError


def track_selection_forward(module, object_tokens, lang_tokens):
    raise SolaError(
        "sola_amd: the differentiable (training) forward is not built yet; call the module under torch.no_grad() "
        "with module.eval() for the forward+loss path. There is no PyTorch fallback.")
