"""Drop-in for the reference's ``module/module.py``: same class name, constructor dict, forward signature,
attributes and state_dict keys, with the numerics running in libsola_hip.so on the MI355X.

``torch`` is used for parameter storage, device memory and the current stream only.  The standard
``nn.Conv1d`` / ``nn.GroupNorm`` / ``nn.Linear`` / ``nn.Embedding`` objects below are parameter HOLDERS
(their ``forward`` is never called); constructing them in the reference's order gives the reference's
state_dict keys and - under the same ``torch.manual_seed`` - the reference's initial weights
(module/module.py:74-110, tools/attention.py:26-29).
"""
from __future__ import annotations

import ctypes as C

import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import SolaConfig, SolaError, SolaRaggedBatch, check, current_stream, lib, ptr, require_cuda

NUM_HEADS = 8  # module/module.py:13-15


class _AttentionParams(nn.Module):
    """Parameter holder with the key layout of tools/attention.py:26-29 (attention dropout 0.1, :12)."""

    def __init__(self, embed_dim, num_heads, dropout_p=0.1):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout_p = embed_dim, num_heads, dropout_p
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)

    def forward(self, *a, **k):  # pragma: no cover
        raise SolaError("parameter holder: the attention runs inside libsola_hip (sola_forward)")


class ObjectLanguageAlignmentLayer(nn.Module):
    """Parameter holder for one alignment layer (module/module.py:8-20)."""

    def __init__(self, configs):
        super().__init__()
        D = configs["lang_token_dim"]
        self.obj_attn = _AttentionParams(D, NUM_HEADS)
        self.motion_attn = _AttentionParams(D, NUM_HEADS)
        self.object2lang_attn = _AttentionParams(D, NUM_HEADS)
        self.norm = nn.ModuleList([nn.GroupNorm(configs["n_groups_module"], D) for _ in range(3)])

    def forward(self, *a, **k):  # pragma: no cover
        raise SolaError("parameter holder: the layer runs inside libsola_hip (sola_forward)")


def _rows_of(tensors, cols):
    """The tensors of a ragged batch as ONE [rows, cols] f32 matrix.  Tensors that already lie back to back in one buffer (a collated batch:
    ``collate_ragged``) are taken as they are - no copy; anything else is concatenated (torch.cat: a pass over the batch's bytes)."""
    t0 = tensors[0]
    ok = all(t.dtype == torch.float32 and t.is_contiguous() and t.device == t0.device for t in tensors)
    if ok:
        end = t0.data_ptr()
        for t in tensors:
            if t.data_ptr() != end or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr():
                ok = False
                break
            end += t.numel() * 4
    if ok:
        rows = sum(int(t.numel()) for t in tensors) // cols
        return torch.as_strided(t0, (rows, cols), (cols, 1), t0.storage_offset())
    return torch.cat([t.reshape(-1, cols) for t in tensors], 0).to(torch.float32).contiguous()


def collate_ragged(tensors, device=None):
    """Views of ONE contiguous buffer holding the given per-sample tensors back to back (what a collate function hands the training loop):
    ``forward_ragged`` / ``train_step_ragged`` then read the buffer where it lies instead of concatenating the batch on the device."""
    flat = torch.cat([t.reshape(-1) for t in tensors]).to(torch.float32)
    if device is not None:
        flat = flat.to(device)
    out, o = [], 0
    for t in tensors:
        n = int(t.numel())
        out.append(flat[o:o + n].view(t.shape))
        o += n
    return out


class LanguageAlignedTrackSelectionModule(nn.Module):
    """``m(object_tokens [B,N,T,d], lang_tokens [B,L,D]) -> (score_map [B,N], score_tokens [B,N,D])``
    (module/module.py:54-162)."""

    def __init__(self, configs) -> None:
        super().__init__()
        self.object_token_dim = configs["object_token_dim"]
        self.lang_token_dim = configs["lang_token_dim"]
        self.n_layers = configs["n_layers"]
        self.max_temporal_length = configs["max_temporal_length"]
        self.n_negative = configs["n_negative"]
        self.n_groups = configs["n_groups"]
        self.n_groups_module = configs["n_groups_module"]
        self.dropout_p = configs.get("dropout_p", 0.2)
        assert configs["norm_type"] == "group", "Weight standardization is only supported with group normalization."
        d, h, D = self.object_token_dim, self.object_token_dim * 2, self.lang_token_dim
        p = self.dropout_p
        geo = [(d, h, 3, 2, 1), (h, h, 3, 2, 1), (h, h, 3, 2, 1), (h, D, 3, 1, 1), (D, D, 3, 1, 1)]
        mods = []
        for cin, cout, k, s, pad in geo:
            mods += [nn.Conv1d(cin, cout, kernel_size=k, stride=s, padding=pad),
                     nn.GroupNorm(self.n_groups, cout), nn.LeakyReLU(), nn.Dropout(p=p)]
        mods.append(nn.Conv1d(D, D, kernel_size=1, stride=1, padding=0))
        self.short_motion_encoder = nn.Sequential(*mods)  # indices 0..20 as in module/module.py:74-96
        self.object_lang_align_layers = nn.ModuleList([ObjectLanguageAlignmentLayer(configs) for _ in range(self.n_layers)])
        self.register_buffer("positional_encoding_gaussian_matrix", torch.randn((1, D // 2)).float())
        self.negative_token = nn.Embedding(self.n_negative, D)
        # library state (not part of the state_dict)
        self._ctx = None
        self._ctx_device = None
        self._bound = {}      # key -> (data_ptr, version)
        self._weights_touched = False
        self._workspace = None
        self._last_shape = None
        self.ws_policy = "auto"
        self.attention_dropout_p = 0.1  # tools/attention.py:12 (hard-coded in the reference)
        # inference arithmetic of the convs / projections: "f32" (exact f32 MFMA) or "f16x3" (split-f16 operands, three
        # f16 MFMAs per product with f32 accumulation, ~22-bit products); in training "f16x3" covers every GEMM of the
        # step, forward and backward (attention and GroupNorm backward stay f32; the weight-gradient sums take plain f16
        # operands - sola_tune "train_dw_f16"); "f16" = 16-bit activation STORAGE for the
        # inference forward, uniform and ragged (plain f16 between kernels, one f16 MFMA per product, f32 accumulate / softmax /
        # statistics; a reduced-precision mode with a stated tolerance); in TRAINING it is mixed
        # precision: every GEMM of the step on plain-f16 casts, one MFMA per product, everything else f32); "bf16" = the same
        # mixed-precision TRAINING step with bfloat16 GEMM operands (BASELINE.json configs[2] "bf16 training" - a
        # build-side mode, the reference itself trains in fp32), inference calls of such a module run "f16x3")
        # Defaults.  INFERENCE calls: "f16x3", the range-guarded split-f16 mode - what bench.py's headline measures, the same 1e-3
        # parity bar and error class as the exact-f32 kernels, backed by those kernels whenever its guard trips.
        # TRAINING calls (differentiable forward + backward): "f32", exact-f32 kernels like the reference (round 4, ADVICE r3: the
        # training step has no range guard, so its reduced-precision forms are opt-in).  Assigning ``module.precision = X`` is the
        # explicit choice and covers both; ``module.train_precision = X`` sets the training side alone (train.py: train.precision /
        # --precision).  SOLA_PRECISION=f32 selects exact f32 everywhere; SOLA_TRAIN_PRECISION sets the training default.
        env = os.environ.get("SOLA_PRECISION")
        self._precision = env or "f16x3"
        self.train_precision = os.environ.get("SOLA_TRAIN_PRECISION") or env or "f32"
        self._ctx_precision = None
        # "f16x3" inference calls are range-guarded: a value outside the split-f16 pairs' range (or GroupNorm weights that
        # would put activations there) makes the library repeat the call on the exact-f32 kernels (one 4-byte read-back and
        # stream sync per call; set False for fully asynchronous calls / graph capture).  See sola_set_split_guard.
        self.split_guard = True
        self._ctx_guard = None
        self._train_ws = None
        self._bwd_ws = None
        self._train_inputs = None
        self._train_shape = None
        self._train_generation = 0
        self._grad_arena = None
        self._grad_ctx = None
        self._grad_bound = None
        self._grads_in_arena = False

    # ------------------------------------------------------------------------------------------ library binding
    def _config_struct(self):
        return SolaConfig(self.object_token_dim, self.lang_token_dim, self.n_layers, self.max_temporal_length,
                          self.n_negative, self.n_groups, self.n_groups_module, NUM_HEADS)

    def _ensure_ctx(self, device):
        if self._ctx is not None and self._ctx_device == device:
            return
        self._release_ctx()
        handle = C.c_void_p()
        cfg = self._config_struct()
        check(lib().sola_ctx_create(C.byref(cfg), device.index if device.index is not None else torch.cuda.current_device(),
                                    C.byref(handle)), "sola_ctx_create")
        self._ctx, self._ctx_device, self._bound = handle, device, {}
        self._ctx_precision = None
        self._ctx_guard = None

    def _release_ctx(self):
        if getattr(self, "_ctx", None) is not None:
            try:
                lib().sola_ctx_destroy(self._ctx)
            finally:
                self._ctx = None

    def __del__(self):
        try:
            self._release_ctx()
        except Exception:
            pass

    # The module's structure is fixed at construction (it mirrors module/module.py:55-110): the parameter list, its names and the
    # state_dict entries are walked ONCE - nn.Module.parameters() re-walks the module tree on every call, ~0.1 ms each, six to
    # eight times per training step (a batch-1 step is 2.8 ms).  The tensors themselves are looked up by owner and name on every
    # use (.to() replaces buffers, a caller may replace a Parameter object).
    def _params(self):
        return [(key, owner._parameters[leaf]) for key, owner, leaf, is_buf in self._state_entries() if not is_buf]

    def _param_list(self):
        return [p for _, p in self._params()]

    def _state_entries(self):
        se = self.__dict__.get("_state_cache")
        if se is None:
            se = []
            for key in self.state_dict(keep_vars=True).keys():
                owner = self
                *path, leaf = key.split(".")
                for part in path:
                    owner = getattr(owner, part)
                se.append((key, owner, leaf, leaf in owner._buffers))
            self.__dict__["_state_cache"] = se
        return se

    @property
    def precision(self):
        """Arithmetic of the inference calls; assigning it is the explicit choice for training calls too (see __init__)."""
        return self._precision

    @precision.setter
    def precision(self, value):
        self._precision = value
        self.train_precision = value

    def weights_changed(self):
        """Tell the library that parameter VALUES changed in place without torch noticing (e.g. an update through ``p.data``
        or a fused optimizer kernel - neither bumps ``Tensor._version``): cached derived copies (standardised conv weights,
        split-f16 / f16 copies of the projection matrices and their scales) are rebuilt on the next call."""
        self._weights_touched = True

    def _bind_weights(self, train=False):
        """Hand the library the current device pointer of every state_dict tensor; flag in-place updates."""
        # A training forward is followed by an optimizer step.  ``Tensor._version`` cannot be trusted to show it:
        # torch.optim.AdamW(fused=True) updates the parameters without bumping it (the split-f16 copies of the projection
        # weights then go stale: an 80-step run diverged to a loss of 70 where exact f32 reached 1.0 - tools/train_converge.py).
        # So every call after a training forward (and after weights_changed()) rebuilds the derived copies.
        changed = bool(getattr(self, "_weights_touched", False))
        self._weights_touched = False
        for key, owner, leaf, is_buf in self._state_entries():
            t = owner._buffers[leaf] if is_buf else owner._parameters[leaf]
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise SolaError(f"{key}: expected a contiguous float32 tensor")
            rec = (t.data_ptr(), t._version)
            old = self._bound.get(key)
            if old is None or old[0] != rec[0]:
                check(lib().sola_set_weight(self._ctx, key.encode(), ptr(t), t.numel()), f"sola_set_weight({key})")
                self._adam_sig = None  # a moved pointer unbinds the optimizer in the library (sola_adamw_bind again)
                changed = True
            elif old[1] != rec[1]:
                changed = True
            self._bound[key] = rec
        if changed:
            check(lib().sola_weights_changed(self._ctx), "sola_weights_changed")
        # "auto": training re-standardises the conv weights every call like module/ws.py, eval caches them until a
        # weight changes; "always" / "cached" force either behaviour (bench.py uses "always")
        every = self.training if self.ws_policy == "auto" else self.ws_policy == "always"
        check(lib().sola_set_ws_policy(self._ctx, 1 if every else 0), "sola_set_ws_policy")
        prec = self.train_precision if train else self._precision
        if prec not in ("f32", "f16x3", "f16", "bf16"):
            raise SolaError(f"precision must be 'f32', 'f16x3', 'f16' or 'bf16', got {prec!r}")
        # "bf16" is a TRAINING mode (bfloat16 GEMM operands, library precision 3); inference calls of a module set to it run the
        # default split-f16 kernels
        want = {"f32": 0, "f16x3": 1, "f16": 2, "bf16": 3 if train else 1}[prec]
        if self._ctx_precision != want:
            check(lib().sola_set_precision(self._ctx, want), "sola_set_precision")
            self._ctx_precision = want
        if self._ctx_guard != bool(self.split_guard):
            check(lib().sola_set_split_guard(self._ctx, 1 if self.split_guard else 0), "sola_set_split_guard")
            self._ctx_guard = bool(self.split_guard)

    def split_fallbacks(self):
        """(calls repeated in exact f32 because the split-f16 range guard tripped, guard bits of the last checked call)."""
        if self._ctx is None:
            return 0, 0
        n, g = C.c_int64(), C.c_int32()
        check(lib().sola_split_fallback_count(self._ctx, C.byref(n), C.byref(g)), "sola_split_fallback_count")
        return int(n.value), int(g.value)

    def _get_workspace(self, nbytes, device):
        if self._workspace is None or self._workspace.numel() < nbytes or self._workspace.device != device:
            self._workspace = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._workspace

    # ------------------------------------------------------------------------------------------ forward
    def _check_inputs(self, object_tokens, lang_tokens):
        if object_tokens.dim() != 4 or lang_tokens.dim() != 3:
            raise SolaError("object_tokens must be [B,N,T,d] and lang_tokens [B,L,D]")
        B, N, T, d = object_tokens.shape
        Bl, L, D = lang_tokens.shape
        if d != self.object_token_dim or D != self.lang_token_dim or Bl != B:
            raise SolaError(f"shape mismatch: object_tokens {tuple(object_tokens.shape)}, lang_tokens {tuple(lang_tokens.shape)}")
        if min(B, N, T, L) < 1:
            raise SolaError(f"empty input: object_tokens {tuple(object_tokens.shape)}, lang_tokens {tuple(lang_tokens.shape)}")

    def forward(self, object_tokens, lang_tokens):
        require_cuda(object_tokens, lang_tokens)
        self._check_inputs(object_tokens, lang_tokens)
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self._param_list())):
            from .autograd import track_selection_forward  # backward kernels
            return track_selection_forward(self, object_tokens, lang_tokens)
        return self._forward_impl(object_tokens, lang_tokens)

    def _forward_impl(self, object_tokens, lang_tokens):
        self._check_inputs(object_tokens, lang_tokens)
        B, N, T, d = object_tokens.shape
        Bl, L, D = lang_tokens.shape
        dev = object_tokens.device
        obj = object_tokens.to(torch.float32).contiguous()
        lang = lang_tokens.to(torch.float32).contiguous()
        self._ensure_ctx(dev)
        self._bind_weights()
        nbytes = lib().sola_workspace_bytes(self._ctx, B, N, T, L)
        ws = self._get_workspace(nbytes, dev)
        score_map = torch.empty((B, N), device=dev, dtype=torch.float32)
        score_tokens = torch.empty((B, N, D), device=dev, dtype=torch.float32)
        check(lib().sola_forward(self._ctx, ptr(obj), ptr(lang), B, N, T, L, ptr(score_map), ptr(score_tokens),
                                 ptr(ws), ws.numel(), current_stream(dev)), "sola_forward")
        self._last_shape = (B, N, T, L)
        return score_map, score_tokens

    # ------------------------------------------------------------------------------------------ ragged batches
    def _ragged_inputs(self, object_tokens, lang_tokens, sample_video):
        V, S = len(object_tokens), len(lang_tokens)
        if V < 1 or S < 1:
            raise SolaError("forward_ragged: need at least one video and one sample")
        if sample_video is None:
            if S != V:
                raise SolaError("forward_ragged: sample_video is required when the sample count differs from the video count")
            sample_video = list(range(S))
        if len(sample_video) != S:
            raise SolaError("forward_ragged: sample_video must have one entry per sample")
        require_cuda(*object_tokens, *lang_tokens)
        d, D = self.object_token_dim, self.lang_token_dim
        for t in object_tokens:
            if t.dim() != 3 or t.shape[2] != d or t.shape[0] < 1 or t.shape[1] < 1:
                raise SolaError(f"forward_ragged: object tokens must be [N,T,{d}], got {tuple(t.shape)}")
        for t in lang_tokens:
            if t.dim() != 2 or t.shape[1] != D or t.shape[0] < 1:
                raise SolaError(f"forward_ragged: text tokens must be [L,{D}], got {tuple(t.shape)}")
        for v in sample_video:
            if not 0 <= int(v) < V:
                raise SolaError(f"forward_ragged: sample_video entry {v} out of range")
        return [int(v) for v in sample_video]

    @staticmethod
    def _ragged_batch_struct(object_tokens, lang_tokens, sample_video):
        V, S = len(object_tokens), len(lang_tokens)
        vN = (C.c_int32 * V)(*[int(t.shape[0]) for t in object_tokens])
        vT = (C.c_int32 * V)(*[int(t.shape[1]) for t in object_tokens])
        sV = (C.c_int32 * S)(*[int(v) for v in sample_video])
        sL = (C.c_int32 * S)(*[int(t.shape[0]) for t in lang_tokens])
        batch = SolaRaggedBatch(V, vN, vT, S, sV, sL)
        batch._keep = (vN, vT, sV, sL)  # the struct only holds pointers
        return batch

    def forward_ragged(self, object_tokens, lang_tokens, sample_video=None, differentiable=None):
        """Score many (video, expression) samples of different shapes in ONE pass.

        object_tokens : list of V tensors [N_v, T_v, d] - one per VIDEO (object set)
        lang_tokens   : list of S tensors [L_i, D]      - one per SAMPLE
        sample_video  : list of S video indices (default: sample i scores video i, S == V)

        Inference (``eval()`` mode or ``torch.no_grad()``; ``differentiable=False``; sola_forward_ragged): everything
        that does not depend on the text (encoder, layer 0's inter-object and motion sub-blocks) runs once per video and is
        shared by the samples that refer to it; inference.py:44-58 recomputes it per expression.
        Training (``train()`` mode with grad enabled, or ``differentiable=True`` - e.g. gradients in eval mode, dropout off;
        sola_forward_train_ragged / sola_backward_ragged): the whole training step of train.py:62-137 over
        a batch of variable-shape samples; every sample runs its own encoder pass under its own dropout masks (a video referred
        to by several samples is repeated), the returned tensors are differentiable with respect to the parameters.
        Returns ``(score_maps, score_tokens)``: lists of S tensors [N_i] and [N_i, D] (views of two flat buffers, also
        available as ``self.last_ragged``: flat score_map, flat score_tokens, int32 track offsets on the device, counts).
        Each sample's result equals ``self(obj[None], lang[None])`` up to f32 summation order."""
        sample_video = self._ragged_inputs(object_tokens, lang_tokens, sample_video)
        if differentiable is None:
            differentiable = torch.is_grad_enabled() and self.training
        if differentiable:
            if not torch.is_grad_enabled():
                raise SolaError("forward_ragged(differentiable=True) under torch.no_grad()")
            from .autograd import track_selection_forward_ragged  # backward kernels
            objs = [object_tokens[v] for v in sample_video]
            score_map, score_tokens = track_selection_forward_ragged(self, objs, list(lang_tokens))
            counts = self.last_ragged[3]
            self.last_ragged = (score_map, score_tokens, self.last_ragged[2], counts)
            return list(torch.split(score_map, counts)), list(torch.split(score_tokens, counts))
        with torch.no_grad():
            return self._forward_ragged_impl(object_tokens, lang_tokens, sample_video)

    def _forward_ragged_impl(self, object_tokens, lang_tokens, sample_video):
        S = len(lang_tokens)
        d, D = self.object_token_dim, self.lang_token_dim
        dev = object_tokens[0].device
        obj = _rows_of(object_tokens, d)
        lang = _rows_of(list(lang_tokens), D)
        batch = self._ragged_batch_struct(object_tokens, lang_tokens, sample_video)
        self._ensure_ctx(dev)
        self._bind_weights()
        nbytes = lib().sola_ragged_workspace_bytes(self._ctx, C.byref(batch))
        if nbytes == 0:
            raise SolaError("forward_ragged: invalid batch description: " + (lib().sola_last_error() or b"").decode())
        ws = self._get_workspace(nbytes, dev)
        counts = [int(object_tokens[int(v)].shape[0]) for v in sample_video]
        total = sum(counts)
        score_map = torch.empty(total, device=dev, dtype=torch.float32)
        score_tokens = torch.empty((total, D), device=dev, dtype=torch.float32)
        check(lib().sola_forward_ragged(self._ctx, ptr(obj), ptr(lang), C.byref(batch), ptr(score_map), ptr(score_tokens),
                                        ptr(ws), ws.numel(), current_stream(dev)), "sola_forward_ragged")
        offs = [0]
        for n in counts:
            offs.append(offs[-1] + n)
        self.last_ragged = (score_map, score_tokens, torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True), counts)
        return list(torch.split(score_map, counts)), list(torch.split(score_tokens, counts))

    # ------------------------------------------------------------------------------------------ training path
    def _size_x16_arena(self, dev):
        """Reduced-precision training modes: the forward keeps its 16-bit operand casts for the backward's dW products in an arena this
        module owns as a torch tensor (sola_set_x16_arena; ``x16_arena_max_bytes`` caps it, 0 = keep nothing).  Sized before a forward
        from what the previous one asked for, + 1/8 (ragged batches differ from step to step); never shrinks below the largest need seen
        unless ``release_x16_arena()`` is called."""
        if self.train_precision == "f32":
            return
        need, cap, used = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        check(lib().sola_x16_arena_info(self._ctx, C.byref(need), C.byref(cap), C.byref(used)), "sola_x16_arena_info")
        limit = getattr(self, "x16_arena_max_bytes", None)
        want = need.value + need.value // 8
        if limit is not None:
            want = min(want, int(limit))
        arena = getattr(self, "_x16_arena", None)
        if arena is not None and (arena.device != dev or (arena.numel() < want)):
            arena = None
        if arena is not None and cap.value != arena.numel():  # a new context (device change): hand it the arena again
            check(lib().sola_set_x16_arena(self._ctx, ptr(arena), arena.numel()), "sola_set_x16_arena")
        if arena is None and want > 0 and not getattr(self, "_x16_arena_failed", False):
            self._x16_arena = None  # release before allocating the larger one
            check(lib().sola_set_x16_arena(self._ctx, None, 0), "sola_set_x16_arena")
            try:
                arena = torch.empty(int(want), dtype=torch.uint8, device=dev)
            except torch.OutOfMemoryError:  # a convenience, not a requirement: the backward casts again
                self._x16_arena_failed = True
                arena = None
            self._x16_arena = arena
            check(lib().sola_set_x16_arena(self._ctx, ptr(arena), 0 if arena is None else arena.numel()), "sola_set_x16_arena")

    def x16_arena_bytes(self):
        """Bytes currently held for kept operand casts (0 in exact-f32 training)."""
        a = getattr(self, "_x16_arena", None)
        return 0 if a is None else int(a.numel())

    def release_x16_arena(self):
        """Give the operand-cast arena back to torch's allocator (call between a backward and the next forward)."""
        if getattr(self, "_ctx", None):
            check(lib().sola_set_x16_arena(self._ctx, None, 0), "sola_set_x16_arena")
        self._x16_arena = None
        self._x16_arena_failed = False

    def _forward_train_impl(self, object_tokens, lang_tokens):
        """sola_forward_train: same numerics as the inference forward, activations kept for sola_backward."""
        B, N, T, d = object_tokens.shape
        _, L, D = lang_tokens.shape
        dev = object_tokens.device
        obj = object_tokens.detach().to(torch.float32).contiguous()
        lang = lang_tokens.detach().to(torch.float32).contiguous()
        self._ensure_ctx(dev)
        self._bind_weights(train=True)
        self._weights_touched = True  # the caller is about to update the parameters (see _bind_weights)
        self._set_step_dropout()
        self._size_x16_arena(dev)
        nbytes = lib().sola_train_workspace_bytes(self._ctx, B, N, T, L)
        if self._train_ws is None or self._train_ws.numel() < nbytes or self._train_ws.device != dev:
            self._train_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        score_map = torch.empty((B, N), device=dev, dtype=torch.float32)
        score_tokens = torch.empty((B, N, D), device=dev, dtype=torch.float32)
        check(lib().sola_forward_train(self._ctx, ptr(obj), ptr(lang), B, N, T, L, ptr(score_map), ptr(score_tokens),
                                       ptr(self._train_ws), self._train_ws.numel(), current_stream(dev)), "sola_forward_train")
        self._train_inputs = (obj, lang)  # the conv0 weight gradient re-reads the tokens
        self._train_shape = (B, N, T, L)
        self._train_generation += 1
        self._workspace = self._train_ws  # workspace_tap reads the arena of the last forward
        return score_map, score_tokens

    def _set_step_dropout(self):
        if self.training and (self.dropout_p > 0 or self.attention_dropout_p > 0):
            # a fresh mask seed per step from torch's (seedable) CPU generator, like nn.Dropout under set_seed(42)
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            check(lib().sola_set_dropout(self._ctx, float(self.dropout_p), float(self.attention_dropout_p), seed), "sola_set_dropout")
        else:
            seed = 0
            check(lib().sola_set_dropout(self._ctx, 0.0, 0.0, 0), "sola_set_dropout")
        self._last_dropout_seed = seed

    def _forward_train_ragged_impl(self, object_tokens, lang_tokens):
        """sola_forward_train_ragged: one sample per entry of the two lists ([N_i, T_i, d], [L_i, D]); activations and the
        unit tables are kept in the workspace for sola_backward_ragged."""
        S = len(lang_tokens)
        d, D = self.object_token_dim, self.lang_token_dim
        dev = object_tokens[0].device
        obj = _rows_of([t.detach() for t in object_tokens], d)
        lang = _rows_of([t.detach() for t in lang_tokens], D)
        batch = self._ragged_batch_struct(object_tokens, lang_tokens, list(range(S)))
        self._ensure_ctx(dev)
        self._bind_weights(train=True)
        self._weights_touched = True  # the caller is about to update the parameters (see _bind_weights)
        self._set_step_dropout()
        self._size_x16_arena(dev)
        nbytes = lib().sola_train_ragged_workspace_bytes(self._ctx, C.byref(batch))
        if nbytes == 0:
            raise SolaError("forward_ragged (training): invalid batch description: " + (lib().sola_last_error() or b"").decode())
        if self._train_ws is None or self._train_ws.numel() < nbytes or self._train_ws.device != dev:
            self._train_ws = None  # release before allocating the larger arena
            self._train_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        counts = [int(t.shape[0]) for t in object_tokens]
        total = sum(counts)
        score_map = torch.empty(total, device=dev, dtype=torch.float32)
        score_tokens = torch.empty((total, D), device=dev, dtype=torch.float32)
        check(lib().sola_forward_train_ragged(self._ctx, ptr(obj), ptr(lang), C.byref(batch), ptr(score_map), ptr(score_tokens),
                                              ptr(self._train_ws), self._train_ws.numel(), current_stream(dev)),
              "sola_forward_train_ragged")
        self._train_inputs = (obj, lang)  # the conv0 weight gradient re-reads the tokens
        self._train_shape = ("ragged", batch, total)
        self._train_generation += 1
        self._workspace = self._train_ws
        offs = [0]
        for n in counts:
            offs.append(offs[-1] + n)
        self.last_ragged = (score_map, score_tokens, torch.tensor(offs, dtype=torch.int32).to(dev, non_blocking=True), counts)
        return score_map, score_tokens

    def _grad_layout(self):
        """Flat gradient arena: every parameter's gradient is a view of ONE allocation, laid out bucket by bucket in the
        order sola_backward finishes them (sola_grad_bucket_of), so the multi-GPU path all-reduces each bucket in place
        while the rest of the backward still runs (sola_amd/dist.py: allreduce_gradient_arena).  64-float alignment keeps
        every view 256-byte aligned."""
        named = self._params()
        dev = named[0][1].device
        if getattr(self, "_grad_arena", None) is not None and self._grad_arena.device == dev and self._grad_ctx is self._ctx:
            return
        nb = lib().sola_grad_bucket_count(self._ctx)
        # Parameters that ALSO receive gradient from outside the network go behind the last bucket, outside every bucket's flat
        # range: negative_token.weight feeds the alignment loss directly (train.py:92), so autograd ADDS the loss's gradient to
        # this slot on the caller's stream after the whole backward - an in-place collective on a side stream must never cover
        # the slot (it raced with that add: ranks could disagree on whether the add saw the slot before or after the collective,
        # and every rank then ended with the same WRONG gradient).  The tail is reduced on the caller's stream
        # (sola_amd/dist.py: allreduce_gradient_arena).
        tail_keys = {"negative_token.weight"}
        order = []
        for key, p in named:
            b = lib().sola_grad_bucket_of(self._ctx, key.encode())
            if b < 0:
                check(b, f"sola_grad_bucket_of({key})")
            order.append((nb if key in tail_keys else b, len(order), key, p))
        order.sort(key=lambda t: (t[0], t[1]))
        off, spans, starts = 0, [], {}
        for b, _i, key, p in order:
            starts.setdefault(b, off)
            spans.append((key, off, p.numel(), tuple(p.shape)))
            off += (p.numel() + 63) // 64 * 64
        self._grad_arena = torch.zeros(off, device=dev, dtype=torch.float32)
        self._grad_spans = {key: (o, n, shape) for key, o, n, shape in spans}
        bounds = [starts[b] for b in range(nb)] + [starts.get(nb, off)]
        self._grad_buckets = [(bounds[b], bounds[b + 1]) for b in range(nb)]
        self._grad_tail = sorted(tail_keys & set(self._grad_spans))
        self._bind_grad_arena()
        self._grad_ctx = self._ctx

    def _grad_view(self, key):
        """A NEW tensor object over the parameter's slot of the arena.  autograd's AccumulateGrad adopts an incoming gradient
        without copying only if nobody else holds that tensor object, so the views are made per backward and not kept."""
        o, n, shape = self._grad_spans[key]
        return self._grad_arena[o:o + n].view(shape)

    def _bind_grad_arena(self):
        for key in self._grad_spans:
            g = self._grad_view(key)
            check(lib().sola_set_grad(self._ctx, key.encode(), ptr(g), g.numel()), f"sola_set_grad({key})")
        self._grad_bound = "arena"
        self._adam_sig = None  # a moved gradient pointer unbinds the optimizer in the library

    def grad_buckets(self):
        """[(flat view of bucket k)] of the gradient arena, in completion order (valid after a backward)."""
        return [self._grad_arena[a:b] for a, b in self._grad_buckets]

    def _backward_impl(self, d_score_map, d_score_tokens):
        ragged = self._train_shape[0] == "ragged"
        dev = self._train_ws.device
        D = self.lang_token_dim
        if ragged:
            _tag, batch, total = self._train_shape
            sm_shape, st_shape = (total,), (total, D)
        else:
            B, N, T, L = self._train_shape
            sm_shape, st_shape = (B, N), (B, N, D)
        d_sm = torch.zeros(sm_shape, device=dev) if d_score_map is None else d_score_map.to(torch.float32).contiguous()
        d_st = torch.zeros(st_shape, device=dev) if d_score_tokens is None else d_score_tokens.to(torch.float32).contiguous()
        named = dict(self._params())
        self._grad_layout()
        lo, hi = self._grad_arena.data_ptr(), self._grad_arena.data_ptr() + 4 * self._grad_arena.numel()
        # sola_backward OVERWRITES its gradient buffers.  If a parameter still holds a gradient that lives in the arena (the
        # caller accumulates over several backwards instead of zero_grad(set_to_none=True)), writing there would clobber
        # what autograd is about to add to: such a step gets fresh buffers, exactly the pre-arena behaviour.
        aliased = any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in named.values())
        if aliased:
            grads = []
            for key, p in named.items():
                g = torch.empty_like(p, memory_format=torch.contiguous_format)
                check(lib().sola_set_grad(self._ctx, key.encode(), ptr(g), g.numel()), f"sola_set_grad({key})")
                grads.append(g)
            self._grad_bound = "fresh"
            self._adam_sig = None
        else:
            if self._grad_bound != "arena":
                self._bind_grad_arena()
            grads = [self._grad_view(key) for key in named]
        if ragged:
            nbytes = lib().sola_backward_ragged_workspace_bytes(self._ctx, C.byref(batch))
        else:
            nbytes = lib().sola_backward_workspace_bytes(self._ctx, B, N, T, L)
        if self._bwd_ws is None or self._bwd_ws.numel() < nbytes or self._bwd_ws.device != dev:
            self._bwd_ws = None
            self._bwd_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        fn = lib().sola_backward_ragged if ragged else lib().sola_backward
        check(fn(self._ctx, ptr(d_sm), ptr(d_st), ptr(self._train_ws), ptr(self._bwd_ws), self._bwd_ws.numel(),
                 current_stream(dev)), "sola_backward_ragged" if ragged else "sola_backward")
        self._grads_in_arena = not aliased
        return grads

    def train_step(self, object_tokens, lang_tokens, labels, pos_tokens, positive_weight=1.5, temperature=0.07, alignment_weight=0.3,
                   max_grad_norm=0.0, optimizer=None, write_back_grads=True):
        """The body of the reference's training loop (train.py:62-125: forward, weighted BCE + alignment loss on the module's own negative
        tokens, ``loss.backward()``, ``get_grad_norm_dict()``, gradient clipping) as ONE library call (sola_train_step): the ~110 launches
        of a one-sample step are enqueued from C++ instead of call by call through autograd and ctypes (1.6-2.4 ms of host time per step,
        more than their GPU time).  Same kernels in the same order as ``module(...)`` + ``track_selection_losses`` + ``.backward()`` +
        ``clip_grad_norm_``: bit-identical gradients.  Uniform batches ``object_tokens [B,N,T,d]``, ``lang_tokens [B,L,D]``,
        ``labels [B,N]``, ``pos_tokens [B,1,D]``; exact-f32 or any training precision (``train_precision``).

        Returns ``(loss3, score_map, score_tokens)`` - ``loss3 = [total, bce, alignment]`` on the device, no host sync.  Every parameter's
        ``.grad`` is (a view of) the gradient arena afterwards, so ``optimizer.step()`` follows directly; no ``zero_grad`` is needed
        (the backward overwrites).  ``max_grad_norm <= 0``: no clipping (multi-GPU: all-reduce first, then ``clip_grad_norm_``).
        ``get_grad_norm_dict()`` after the call reads the step's own reduction (one host sync).

        ``optimizer``: a ``torch.optim.AdamW`` over exactly this module's parameters (one param group, no amsgrad / maximize) - the clipping and
        the update then run as ONE more launch here (sola_adamw_step: torch's fused AdamW arithmetic, bit-identical parameters and moments; the
        optimizer's own state tensors are updated, ``optimizer.step()`` must NOT be called for this step).``write_back_grads=False`` (with ``optimizer``): an active clip leaves ``.grad`` unclipped instead of rewriting it - the parameters,
        the moments and the reported norms are the same; an eighth less traffic in the update (train.py:121-125 never reads ``.grad`` again).
        """
        require_cuda(object_tokens, lang_tokens, labels, pos_tokens)
        self._check_inputs(object_tokens, lang_tokens)
        B, N, T, _d = object_tokens.shape
        _, L, D = lang_tokens.shape
        dev = object_tokens.device
        f = lambda t: t.detach().to(torch.float32).contiguous()
        obj, lang, lab, pos = f(object_tokens), f(lang_tokens), f(labels), f(pos_tokens)
        if lab.numel() != B * N or pos.numel() != B * D:
            raise SolaError("train_step: labels must be [B,N] and pos_tokens [B,1,D]")
        self._ensure_ctx(dev)
        self._bind_weights(train=True)
        self._weights_touched = True  # the caller is about to update the parameters
        self._set_step_dropout()
        self._size_x16_arena(dev)
        named = self._params()
        self._grad_layout()
        if self._grad_bound != "arena":
            self._bind_grad_arena()
        if getattr(self, "_step_bound_ctx", None) is not self._ctx:
            groups = self._grad_groups()
            key_of = {id(p): k for k, p in named}
            names, gids = [], []
            for gi, (_name, params) in enumerate(groups):
                for p in params:
                    names.append(key_of[id(p)].encode())
                    gids.append(gi)
            arr = (C.c_char_p * len(names))(*names)
            check(lib().sola_train_step_bind(self._ctx, arr, (C.c_int32 * len(gids))(*gids), len(names), len(groups)), "sola_train_step_bind")
            self._step_bound_ctx = self._ctx
        nb_t = lib().sola_train_workspace_bytes(self._ctx, B, N, T, L)
        if self._train_ws is None or self._train_ws.numel() < nb_t or self._train_ws.device != dev:
            self._train_ws = torch.empty(int(nb_t), dtype=torch.uint8, device=dev)
        nb_b = lib().sola_backward_workspace_bytes(self._ctx, B, N, T, L)
        if self._bwd_ws is None or self._bwd_ws.numel() < nb_b or self._bwd_ws.device != dev:
            self._bwd_ws = None
            self._bwd_ws = torch.empty(int(nb_b), dtype=torch.uint8, device=dev)
        nb_s = lib().sola_train_step_workspace_bytes(self._ctx, B, N)
        sw = getattr(self, "_step_ws", None)
        if sw is None or sw.numel() < nb_s or sw.device != dev:
            self._step_ws = sw = torch.empty(int(nb_s), dtype=torch.uint8, device=dev)
        n_groups = len(self._grad_groups())
        score_map = torch.empty((B, N), device=dev, dtype=torch.float32)
        score_tokens = torch.empty((B, N, D), device=dev, dtype=torch.float32)
        loss3 = torch.empty(3, device=dev, dtype=torch.float32)
        grad_sq = torch.empty(n_groups + 1, device=dev, dtype=torch.float64)
        adam = self._adamw_plan(optimizer) if optimizer is not None else None
        check(lib().sola_train_step(self._ctx, ptr(obj), ptr(lang), B, N, T, L, ptr(lab), ptr(pos), float(positive_weight), float(temperature),
                                    float(alignment_weight), 0.0 if adam is not None else float(max_grad_norm), ptr(score_map), ptr(score_tokens), ptr(loss3), ptr(grad_sq),
                                    ptr(self._train_ws), self._train_ws.numel(), ptr(self._bwd_ws), self._bwd_ws.numel(), ptr(sw), sw.numel(),
                                    current_stream(dev)), "sola_train_step")
        self._train_inputs = (obj, lang)
        self._train_shape = (B, N, T, L)
        self._train_generation += 1
        self._workspace = self._train_ws
        self._grads_in_arena = True
        self._step_grad_sq = grad_sq  # sums of squares of the UNclipped gradients (module/module.py:164-199 reports their roots)
        self._last_grad_sq = None
        if adam is not None:  # clip + AdamW in one launch; the kernel reads the total norm where the step left it
            g = optimizer.param_groups[0]
            # step 0: the update's number comes from the optimizer's own device step tensors (torch's optimizer.step() may be mixed in)
            check(lib().sola_adamw_step(self._ctx, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                        0, C.c_void_p(grad_sq.data_ptr() + 8 * n_groups), float(max_grad_norm), 1 if write_back_grads else 0,
                                        current_stream(dev)), "sola_adamw_step")
        # every parameter's .grad = its slot of the arena (persistent views: nothing to do from the second step on)
        lo, hi = self._grad_arena.data_ptr(), self._grad_arena.data_ptr() + 4 * self._grad_arena.numel()
        for key, p in named:
            g = p.grad
            if g is None or not (lo <= g.data_ptr() < hi):
                p.grad = self._grad_view(key)
        return loss3, score_map, score_tokens

    def optimizer_step(self, optimizer, max_grad_norm=0.0, write_back_grads=True):
        """``clip_grad_norm_(max_grad_norm)`` + ``optimizer.step()`` (train.py:121-125) as ONE launch for a ``torch.optim.AdamW`` over this
        module's parameters: the gradients' norm is reduced on the device (or taken from a ``get_grad_norm_dict()`` of these very gradients),
        the clip decision, the scaling and the update happen in sola_adamw_step with torch's fused arithmetic (bit-identical parameters and
        moments; the optimizer's own state tensors are updated).  Call it where the loop called ``clip_grad_norm_`` and ``optimizer.step()``
        - after the backward and, with several ranks, after the gradient all-reduce.  Needs the gradients in the module's arena (a
        ``zero_grad(set_to_none=True)`` loop, the default).  ``write_back_grads=False``: an active clip does not rewrite ``.grad`` (it keeps
        the unclipped gradient; parameters and moments are the same) - an eighth less traffic for loops that, like train.py:121-125, never
        read the gradients behind the step."""
        named = self._params()
        if any(p.grad is None for _, p in named) or not getattr(self, "_grads_in_arena", False):
            raise SolaError("optimizer_step: every parameter needs a gradient in the module's arena (run the backward after zero_grad(set_to_none=True))")
        base = self._grad_arena.data_ptr()
        for key, p in named:  # a gradient autograd accumulated into a tensor of its own (negative_token.weight: two contributions) goes back to its slot
            o, _n, _shape = self._grad_spans[key]
            if p.grad.data_ptr() != base + 4 * o:
                view = self._grad_view(key)
                view.copy_(p.grad)
                p.grad = view
        grads = [p.grad for _, p in named]
        dev = grads[0].device
        self._adamw_plan(optimizer)
        total = None
        if max_grad_norm and max_grad_norm > 0:
            cached = getattr(self, "_last_grad_sq", None)
            sq = cached[0] if cached is not None and cached[1] == self._grad_tag(grads) else self._grad_sq_device()
            total = sq[-1:]
        g = optimizer.param_groups[0]
        check(lib().sola_adamw_step(self._ctx, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                                    0, ptr(total), float(max_grad_norm or 0.0), 1 if write_back_grads else 0, current_stream(dev)), "sola_adamw_step")
        self._last_grad_sq = None
        self._weights_touched = True

    def _adamw_plan(self, opt):
        """Bind a torch.optim.AdamW's state tensors to the context (sola_adamw_bind) - once, and again when a pointer changed."""
        named = self._params()
        if type(opt) is not torch.optim.AdamW or len(opt.param_groups) != 1:
            raise SolaError("train_step(optimizer=...): a torch.optim.AdamW with one parameter group")
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or not isinstance(g["lr"], float):
            raise SolaError("train_step(optimizer=...): amsgrad / maximize / capturable / differentiable / tensor learning rates are not supported")
        if [id(p) for p in g["params"]] != [id(p) for _, p in named]:
            raise SolaError("train_step(optimizer=...): the optimizer must hold exactly this module's parameters, in parameters() order")
        for _, p in named:  # torch creates the state at its first step(): the same tensors, the same way (Adam._init_group, fused)
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        sig = (id(opt), id(self._ctx), self._grad_bound) + tuple((p.data_ptr(), opt.state[p]["exp_avg"].data_ptr(), opt.state[p]["exp_avg_sq"].data_ptr(),
                                                                  opt.state[p]["step"].data_ptr()) for _, p in named)
        if getattr(self, "_adam_sig", None) != sig:
            steps = [opt.state[p]["step"] for _, p in named]
            if any(not s_.is_cuda or s_.dtype != torch.float32 for s_ in steps):
                raise SolaError("train_step(optimizer=...): the optimizer's step counters must be device float32 scalars (torch.optim.AdamW(fused=True) or fresh state)")
            n = len(named)
            names = (C.c_char_p * n)(*[k.encode() for k, _ in named])
            vp = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            check(lib().sola_adamw_bind(self._ctx, names, vp([opt.state[p]["exp_avg"] for _, p in named]), vp([opt.state[p]["exp_avg_sq"] for _, p in named]),
                                        vp(steps), n), "sola_adamw_bind")
            self._adam_sig = sig
        return True

    def step_grad_norm_dict(self):
        """``get_grad_norm_dict()`` (module/module.py:164-199) of the last ``train_step``: the norms of its gradients BEFORE clipping, from
        the reduction the step ran on the device (one host sync)."""
        sq = getattr(self, "_step_grad_sq", None)
        if sq is None:
            raise SolaError("step_grad_norm_dict: no train_step yet")
        groups = self._grad_groups()
        vals = sq[:len(groups)].cpu().tolist()
        out = {"total_grad_norm": sum(vals) ** 0.5}
        for (name, _), v in zip(groups, vals):
            out[name] = v ** 0.5
        return out

    def workspace_tap(self, name):
        """Copy of a named intermediate of the last forward (see sola_workspace_tap); for parity tests."""
        off, rows, cols = C.c_size_t(), C.c_int64(), C.c_int64()
        check(lib().sola_workspace_tap(self._ctx, name.encode(), C.byref(off), C.byref(rows), C.byref(cols)), "sola_workspace_tap")
        n = rows.value * cols.value
        flat = self._workspace[off.value: off.value + 4 * n].view(torch.float32)
        return flat.reshape(rows.value, cols.value).clone()

    # ------------------------------------------------------------------------------------------ a7
    def _grad_groups(self):
        groups = self.__dict__.get("_group_cache")
        if groups is None:
            groups = [("short_motion_encoder", list(self.short_motion_encoder.parameters()))]
            groups += [(f"scmola_layer_{i}", list(layer.parameters())) for i, layer in enumerate(self.object_lang_align_layers)]
            groups.append(("negative_token", list(self.negative_token.parameters())))
            self.__dict__["_group_cache"] = groups
        return groups

    def _grad_sq_device(self):
        """Per-group sums of squares of the gradients + their total as device doubles (one multi-tensor launch, NO host
        sync); None when there are no gradients."""
        groups = self._grad_groups()
        tensors, group_ids = [], []
        for gi, (_name, params) in enumerate(groups):
            for p in params:
                if p.grad is not None:
                    tensors.append(p.grad)
                    group_ids.append(gi)
        if not tensors:
            return None
        return self._grad_sqnorms(tensors, group_ids, len(groups))

    def get_grad_norm_dict(self):
        """module/module.py:164-199 with one device reduction per group and a single host sync."""
        groups = self._grad_groups()
        n_groups = len(groups)
        sq = self._grad_sq_device()
        if sq is None:
            vals = [0.0] * n_groups
        else:
            # device doubles, reused by clip_grad_norm_ without another reduction - but only for these very gradient
            # values: the tag is every gradient's (storage, version), so an all-reduce, unscale, accumulation or a new
            # backward in between makes clip_grad_norm_ reduce again instead of clipping with a stale norm
            self._last_grad_sq = (sq, self._grad_tag([p.grad for p in self._param_list() if p.grad is not None]))
            vals = sq[:n_groups].cpu().tolist()  # the single host sync
        out = {"total_grad_norm": sum(vals) ** 0.5}
        for (name, _), v in zip(groups, vals):
            out[name] = v ** 0.5
        return out

    def _grad_sqnorms(self, tensors, group_ids, n_groups):
        """sola_grad_sqnorms: per-group sum of squares in one multi-tensor launch; slot n_groups holds the total."""
        require_cuda(*tensors)
        n = len(tensors)
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise SolaError("gradients must be contiguous float32 tensors")
        dev = tensors[0].device
        if n > 128:
            raise SolaError("too many gradient tensors for one launch")
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        numel = (C.c_int64 * n)(*[t.numel() for t in tensors])
        group = (C.c_int32 * n)(*group_ids)
        out = torch.empty(n_groups + 1, dtype=torch.float64, device=dev)  # [groups..., total]
        nb = lib().sola_grad_sqnorms_scratch_bytes(n, numel)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        check(lib().sola_grad_sqnorms(ptrs, numel, group, n, n_groups, ptr(out), ptr(scratch), nb, current_stream(dev)),
              "sola_grad_sqnorms")
        return out

    def _grad_tag(self, grads):
        # the HIP backward writes gradient storage behind torch's back (no version bump): the step counter covers that
        return (self._train_generation,) + tuple((g.data_ptr(), g._version) for g in grads)

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm) (train.py:121-122) as one in-place multi-tensor
        launch: the total norm is reduced on the device (or taken from a get_grad_norm_dict() of these very gradients) and
        the kernel itself decides whether to scale - the reference's ``if total_grad_norm > clip`` without a host sync."""
        grads = [p.grad for p in self._param_list() if p.grad is not None]
        if not grads:
            return
        cached = getattr(self, "_last_grad_sq", None)
        if cached is not None and cached[1] == self._grad_tag(grads):
            sq = cached[0]
        else:
            sq = self._grad_sq_device()  # reduced on the device; the clip kernel reads the total there: no host sync
        n = len(grads)
        dev = grads[0].device
        ptrs = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
        numel = (C.c_int64 * n)(*[g.numel() for g in grads])
        total = sq[-1:]
        check(lib().sola_grad_clip(ptrs, numel, n, ptr(total), float(max_norm), current_stream(dev)), "sola_grad_clip")
        self._last_grad_sq = None
