"""Text side of the path: frozen RoBERTa stays on stock PyTorch-ROCm (north star); this module only wraps it.

``encode(expressions) -> (lang_tokens [B,L,D], pos_tokens [B,1,D])`` is train.py:80-91 (tokenise, last_hidden_state,
attention-mask mean pooling).  A missing checkpoint is an ERROR, as in the reference (train.py:31-35 fails when
``AutoModel.from_pretrained`` does): masks, metrics or weights computed from meaningless language tokens must not be
written with exit code 0.  Neither box here has network access or the checkpoint, so a deterministic hashed embedding
with the same output contract exists for plumbing runs - only when asked for explicitly: ``allow_standin=True``
(the entry points pass it for ``--synthetic true``) or ``SOLA_ALLOW_TEXT_STANDIN=1``.  ``TextEncoder.kind`` says which
encoder produced the tokens ("roberta" / "hashed-standin"); eval.py records it in its JSON."""
from __future__ import annotations

import hashlib
import warnings

import numpy as np
import torch


class TextEncoder:
    def __init__(self, name: str, dim: int, device, allow_standin: bool = False):
        import os

        self.dim, self.device = dim, device
        self.tokenizer = self.model = None
        self.kind = "roberta"
        allow_standin = bool(allow_standin) or os.environ.get("SOLA_ALLOW_TEXT_STANDIN", "0") == "1"
        try:
            os.environ.setdefault("HF_HUB_OFFLINE", "1")
            from transformers import AutoModel, AutoTokenizer

            self.tokenizer = AutoTokenizer.from_pretrained(name, local_files_only=True)
            self.model = AutoModel.from_pretrained(name, local_files_only=True).to(device).eval()
        except Exception as e:  # no local checkpoint
            if not allow_standin:
                raise RuntimeError(
                    f"text encoder '{name}' could not be loaded from the local cache ({type(e).__name__}: {e}). "
                    "Refusing to continue with stand-in embeddings: results would be meaningless. For a plumbing run "
                    "pass --synthetic true or set SOLA_ALLOW_TEXT_STANDIN=1.") from e
            self.tokenizer = self.model = None
            self.kind = "hashed-standin"
            warnings.warn(f"text encoder '{name}' is not available offline ({type(e).__name__}); using hashed stand-in "
                          "embeddings because the stand-in was explicitly allowed")

    @torch.no_grad()
    def encode_ragged(self, expressions):
        """``(list of [L_i, D] token tensors, pos [S, D])`` for a ragged batch: every expression keeps its OWN length (the
        reference encodes one expression at a time, train.py:80-91 at batch_size 1, so it never sees padding; here the
        padded positions of the batched encoder are cut off again with the attention mask)."""
        if self.model is not None:
            enc = self.tokenizer.batch_encode_plus(expressions, padding="longest", return_tensors="pt").to(self.device)
            tok = self.model(**enc).last_hidden_state
            lens = enc["attention_mask"].sum(1).tolist()
            toks = [tok[b, :int(n)] for b, n in enumerate(lens)]
        else:
            tok, _ = self.encode(list(expressions))  # one padded host array + ONE copy; every expression keeps its own length
            toks = [tok[b, :len(e.lower().split()) + 2] for b, e in enumerate(expressions)]
        pos = torch.stack([t.mean(0) for t in toks], 0)
        return toks, pos

    @torch.no_grad()
    def encode(self, expressions):
        if self.model is not None:
            enc = self.tokenizer.batch_encode_plus(expressions, padding="longest", return_tensors="pt").to(self.device)
            out = self.model(**enc)
            tok = out.last_hidden_state
            m = enc["attention_mask"].unsqueeze(-1).expand(tok.size()).float()
            pos = (torch.sum(tok * m, 1) / torch.clamp(m.sum(1), min=1e-9)).unsqueeze(1)
            return tok, pos
        words = [["<s>"] + e.lower().split() + ["</s>"] for e in expressions]
        L = max(len(w) for w in words)
        tok = np.zeros((len(words), L, self.dim), dtype=np.float32)
        mask = np.zeros((len(words), L, 1), dtype=np.float32)
        cache = self.__dict__.setdefault("_word_cache", {})  # a word's stand-in vector is a pure function of the word
        for b, ws in enumerate(words):
            for i, w in enumerate(ws):
                vec = cache.get(w)
                if vec is None:
                    seed = int.from_bytes(hashlib.sha256(w.encode()).digest()[:8], "little")
                    vec = cache[w] = (np.random.Generator(np.random.PCG64(seed)).standard_normal(self.dim) * 0.5).astype(np.float32)
                tok[b, i] = vec
                mask[b, i] = 1.0
        pos = (tok * mask).sum(1, keepdims=True) / np.maximum(mask.sum(1, keepdims=True), 1e-9)
        return torch.from_numpy(tok).to(self.device), torch.from_numpy(pos).to(self.device)
