"""ctypes binding of libsola_hip.so (include/sola_hip.h).

The library is the product: there is NO fallback.  If it is missing or fails to load, importing a
compute entry point raises ``SolaLibraryError`` with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOLA_HIP_LIB") or os.path.join(_HERE, "lib", "libsola_hip.so")  # override: kernel experiments only


class SolaLibraryError(RuntimeError):
    pass


class SolaError(RuntimeError):
    """A libsola_hip call returned a negative status."""


class SolaConfig(C.Structure):
    _fields_ = [
        ("object_token_dim", C.c_int32),
        ("lang_token_dim", C.c_int32),
        ("n_layers", C.c_int32),
        ("max_temporal_length", C.c_int32),
        ("n_negative", C.c_int32),
        ("n_groups", C.c_int32),
        ("n_groups_module", C.c_int32),
        ("num_heads", C.c_int32),
    ]


class SolaRaggedBatch(C.Structure):
    """include/sola_hip.h: host descriptor of a ragged batch (videos = object sets, samples = (video, expression))."""
    _fields_ = [
        ("n_videos", C.c_int32),
        ("video_tracks", C.POINTER(C.c_int32)),
        ("video_frames", C.POINTER(C.c_int32)),
        ("n_samples", C.c_int32),
        ("sample_video", C.POINTER(C.c_int32)),
        ("sample_text_len", C.POINTER(C.c_int32)),
    ]


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/sola_hip.h declares
SIGNATURES = {
    "sola_last_error": (C.c_char_p, []),
    "sola_version": (C.c_char_p, []),
    "sola_ctx_create": (_i, [C.POINTER(SolaConfig), _i, C.POINTER(_vp)]),
    "sola_ctx_destroy": (_i, [_vp]),
    "sola_num_weights": (_i, [_vp]),
    "sola_weight_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64)]),
    "sola_set_weight": (_i, [_vp, C.c_char_p, _vp, _i64]),
    "sola_weights_changed": (_i, [_vp]),
    "sola_set_ws_policy": (_i, [_vp, _i]),
    "sola_set_precision": (_i, [_vp, _i]),
    "sola_set_split_guard": (_i, [_vp, _i]),
    "sola_split_fallback_count": (_i, [_vp, C.POINTER(_i64), C.POINTER(C.c_int32)]),
    "sola_cast_sp16": (_i, [_vp, _i, _vp, _i, _i64, _i, _f, _vp]),
    "sola_gemm_nt_split": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sola_cast_sp16_auto": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp, _vp]),
    "sola_gemm_nt_split_scaled": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "sola_cast_f16": (_i, [_vp, _i, _vp, _i, _i64, _i, _f, _vp, _vp]),
    "sola_gemm_nt_f16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sola_gemm_nn": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sola_gemm_nt_bf16": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "sola_attention_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp, _vp]),
    "sola_attention_backward_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                          _i64, _i64, _i64, _i64, _i64, _i64, _f, _i64, _vp, _sz, _vp]),
    "sola_attention_f16": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "sola_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "sola_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "sola_ragged_workspace_bytes": (_sz, [_vp, C.POINTER(SolaRaggedBatch)]),
    "sola_forward_ragged": (_i, [_vp, _vp, _vp, C.POINTER(SolaRaggedBatch), _vp, _vp, _vp, _sz, _vp]),
    "sola_loss_ragged": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _i, _i64, _i, _i, _f, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "sola_train_ragged_workspace_bytes": (_sz, [_vp, C.POINTER(SolaRaggedBatch)]),
    "sola_backward_ragged_workspace_bytes": (_sz, [_vp, C.POINTER(SolaRaggedBatch)]),
    "sola_forward_train_ragged": (_i, [_vp, _vp, _vp, C.POINTER(SolaRaggedBatch), _vp, _vp, _vp, _sz, _vp]),
    "sola_backward_ragged": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sola_loss_backward_ragged": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _i, _i64, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sola_workspace_tap": (_i, [_vp, C.c_char_p, C.POINTER(_sz), C.POINTER(_i64), C.POINTER(_i64)]),
    "sola_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    "sola_select": (_i, [_vp, _i64, _f, _vp, _vp, _vp]),
    "sola_ws_standardize": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "sola_gemm_nt": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "sola_conv1d_cl": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "sola_group_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i64, _i, _i, _i, _f, _f, _i, _vp]),
    "sola_attention": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp, _vp]),
    "sola_attention_split": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp, _vp]),
    "sola_set_dropout": (_i, [_vp, _f, _f, C.c_uint64]),
    "sola_set_stage_dropout": (_i, [_f, C.c_uint64]),
    "sola_train_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "sola_backward_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "sola_set_grad": (_i, [_vp, C.c_char_p, _vp, _i64]),
    "sola_forward_train": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "sola_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sola_grad_bucket_count": (_i, [_vp]),
    "sola_grad_bucket_of": (_i, [_vp, C.c_char_p]),
    "sola_backward_wait_bucket": (_i, [_vp, _i, _vp]),
    "sola_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "sola_ws_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sola_gemm_tn_scratch_bytes": (_sz, [_i, _i, _i]),
    "sola_gemm_tn_split_scratch_bytes": (_sz, [_i, _i, _i]),
    "sola_gemm_tn_split": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sola_gemm_tn_f16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sola_conv1d_cl_wgrad_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sola_gemm_tn": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "sola_conv1d_cl_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sola_conv1d_cl_backward_split_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "sola_conv1d_cl_backward_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "sola_group_norm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i64, _i64, _i, _i, _i, _f, _f, _i, _vp, _sz, _vp]),
    "sola_attention_backward": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                     _i64, _i64, _i64, _i64, _i64, _i64, _f, _vp]),
    "sola_attention_backward_scratch_floats": (_sz, [_i64, _i, _i, _i]),
    "sola_attention_backward_ws": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                        _i64, _i64, _i64, _i64, _i64, _i64, _f, _i64, _vp, _sz, _vp]),
    "sola_pos_encoding": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "sola_mask_words": (_i64, [_i, _i]),
    "sola_mask_pack": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sola_mask_pair_counts": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "sola_mask_bilinear_pack": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sola_mask_unpack": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "sola_rle_fill_or": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sola_rle_string_to_cum": (_i64, [C.c_char_p, _i64, _vp, _i64, _i64]),
    "sola_mask_iou_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "sola_mask_iou_matrix": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "sola_grad_sqnorms_scratch_bytes": (_sz, [_i, _vp]),
    "sola_grad_sqnorms": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "sola_grad_clip": (_i, [_vp, _vp, _i, _vp, _f, _vp]),
    "sola_set_x16_arena": (_i, [_vp, _vp, _sz]),
    "sola_x16_arena_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "sola_tune": (_i, [C.c_char_p, _i]),
    "sola_has_experiments": (_i, []),
    "sola_selftest": (_i, [_vp]),
    "sola_adamw_bind": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i]),
    "sola_adamw_step": (_i, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i64, _vp, _f, _i, _vp]),
    "sola_train_step_bind": (_i, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_int32), _i, _i]),
    "sola_train_step_workspace_bytes": (_sz, [_vp, _i, _i]),
    "sola_train_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    "sola_gemm_trace_read": (C.c_longlong, [_vp, C.c_longlong]),
    "sola_profile_enable": (_i, [_i]),
    "sola_profile_read": (_i, [C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), _i]),
}

PROF_CATEGORIES = ["gemm128", "attn", "group_norm", "ws_standardize", "head_loss", "misc", "iou_pack", "iou_pair", "gemm64", "gemm_tn", "attn_bwd", "gemm_split",
                   "gemm_split256", "gemm_split256_gn"]

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises SolaLibraryError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the HIP runtime of the process (it owns the device memory and streams
    # handed to the library), so it is loaded before libsola_hip.so resolves its HIP symbols.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise SolaLibraryError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sola_amd/csrc`). sola_amd has no CPU or PyTorch fallback.")
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise SolaLibraryError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise SolaLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    # kernel experiments only: SOLA_TUNE="key=value,key=value" applies sola_tune settings at load time (tools/, never the tests or bench defaults)
    for kv in filter(None, os.environ.get("SOLA_TUNE", "").split(",")):
        k, v = kv.split("=")
        if h.sola_tune(k.strip().encode(), int(v)) != 0:
            raise SolaLibraryError(f"SOLA_TUNE: unknown key {k!r}")
    _lib = h
    return h


def has_experiments():
    """True when the library was built with EXPERIMENTS=1 (closed experiments' kernels and their sola_tune keys compiled in)."""
    return bool(lib().sola_has_experiments())


def check(status, what=""):
    if status != 0:
        msg = lib().sola_last_error()
        raise SolaError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch

    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SolaError("sola_amd runs on the GPU only (tensor on %s); there is no CPU fallback" % t.device)


def profile_enable(on=True, categories=None):
    """``categories``: time only these (names of PROF_CATEGORIES) - every timed launch costs two event records on its stream."""
    v = 1 if on else 0
    if on and categories is not None:
        v = sum(1 << (PROF_CATEGORIES.index(c) + 1) for c in categories)
    check(lib().sola_profile_enable(v), "sola_profile_enable")


def profile_read(reset=True):
    n = len(PROF_CATEGORIES)
    launches = (_i64 * n)()
    ms = (C.c_double * n)()
    flops = (C.c_double * n)()
    nbytes = (C.c_double * n)()
    check(lib().sola_profile_read(launches, ms, flops, nbytes, 1 if reset else 0), "sola_profile_read")
    return {PROF_CATEGORIES[i]: {"launches": int(launches[i]), "ms": float(ms[i]), "flops": float(flops[i]),
                                 "bytes": float(nbytes[i])} for i in range(n)}
