"""Oracle (CPU, torch fp32/fp64) for the T5-style encoder-decoder and the decode loop.
TEST INFRASTRUCTURE (oracle/__init__.py).

Restates   mt3/network.py:25-409  (T5Config, Encoder/Decoder layers, Transformer.encode/decode)
           mt3/layers.py:51-82    (sinusoidal table)        :85-157  (dot_product_attention, UNSCALED)
           mt3/layers.py:164-355  (MultiHeadDotProductAttention + KV cache semantics)
           mt3/layers.py:373-418  (DenseGeneral, bias-free) :435-486 (MlpBlock, gated)
           mt3/layers.py:604-621  (LayerNorm == RMSNorm, eps 1e-6)
           mt3/gin/model.gin:47-59 (hyper-parameters)
and the t5x decode loop the model wrapper selects (mt3/models.py:121-137 ->
t5x `decoding.beam_search`, num_decodes=1) [third-party, from memory]: greedy is
the product semantics; `beam1_decode` emulates t5x's beam-size-1 search.

PINNED for the network itself: encoder output, teacher-forced logits and the cached
one-token decode path match golden vectors produced by the reference's REAL
mt3/layers.py + mt3/network.py, executed unmodified on a numpy stand-in for jax/flax
(tests/golden/make_network_golden.py -> tests/test_oracle_network_golden.py, 2e-5),
plus the attention / mask / DenseGeneral / ReLU-MLP literals of mt3/layers_test.py
(tests/test_oracle_network.py).  `nn.gelu` = tanh approximation [flax.linen.gelu
default approximate=True; cross-checked against torch's].  PARITY UNPINNED: the t5x
decode loop (beam search) -- t5x is neither installable here nor in the reference tree.

Parameters are a flat dict name -> np.float32 array, names/shapes exactly the
Flax tree of SURVEY.md A.3 joined with '/'.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


@dataclasses.dataclass(frozen=True)
class T5Config:
    """network.py:25-41 with the values of model.gin:47-59 as defaults."""
    vocab_size: int = 1536
    emb_dim: int = 512
    num_heads: int = 6
    num_encoder_layers: int = 8
    num_decoder_layers: int = 8
    head_dim: int = 64
    mlp_dim: int = 1024
    input_depth: int = 512
    max_pos: int = 2048          # layers.py:565 FixedEmbed.max_length


def sinusoidal_table(max_len: int, features: int) -> np.ndarray:
    """layers.py:51-82: [sin | cos] halves, scale = -ln(10000)/(features/2 - 1)."""
    pe = np.zeros((max_len, features), np.float32)
    pos = np.arange(max_len)[:, None]
    scale = -np.log(10000.0) / (features // 2 - 1)
    div = np.exp(np.arange(features // 2) * scale)
    pe[:, : features // 2] = np.sin(pos * div)
    pe[:, features // 2: 2 * (features // 2)] = np.cos(pos * div)
    return pe


def rms_norm(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """layers.py:611-621."""
    ms = (x * x).mean(-1, keepdim=True)
    return x * torch.rsqrt(ms + 1e-6) * scale


def attention(q, k, v, bias=None):
    """layers.py:134-157 on [B, len, H, D] tensors; NO 1/sqrt(d) (layers.py:230-234)."""
    w = torch.einsum("bqhd,bkhd->bhqk", q, k)
    if bias is not None:
        w = w + bias
    w = torch.softmax(w, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", w, v)


def mlp_block(x: torch.Tensor, wi: List[torch.Tensor], wo: torch.Tensor, activations=("gelu", "linear")):
    """layers.py:435-486 MlpBlock: product over `activations` of act_i(x @ wi_i), then @ wo."""
    acts = {"relu": torch.relu, "linear": lambda t: t, "gelu": lambda t: F.gelu(t, approximate="tanh")}
    h = None
    for name, w in zip(activations, wi):
        t = acts[name](x @ w)
        h = t if h is None else h * t
    return h @ wo


class Oracle:
    def __init__(self, params: Dict[str, np.ndarray], cfg: T5Config, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.p = {k: torch.from_numpy(np.asarray(v, np.float32)).to(dtype) for k, v in params.items()}
        self.pe = torch.from_numpy(sinusoidal_table(cfg.max_pos, cfg.emb_dim)).to(dtype)

    # ---- building blocks
    def _heads(self, x, w):
        return (x @ w).reshape(*x.shape[:-1], self.cfg.num_heads, self.cfg.head_dim)

    def _mha(self, prefix, xq, xkv, bias=None):
        p = self.p
        q = self._heads(xq, p[prefix + "/query/kernel"])
        k = self._heads(xkv, p[prefix + "/key/kernel"])
        v = self._heads(xkv, p[prefix + "/value/kernel"])
        o = attention(q, k, v, bias)
        return o.reshape(*o.shape[:-2], -1) @ p[prefix + "/out/kernel"]

    def _mlp(self, prefix, x):
        p = self.p
        g = F.gelu(x @ p[prefix + "/wi_0/kernel"], approximate="tanh")
        return (g * (x @ p[prefix + "/wi_1/kernel"])) @ p[prefix + "/wo/kernel"]

    # ---- encoder: network.py:158-193, 275-301 (all-ones mask)
    def encode(self, inputs: np.ndarray) -> torch.Tensor:
        p, cfg = self.p, self.cfg
        x = torch.from_numpy(np.asarray(inputs, np.float32)).to(self.dtype)
        T = x.shape[-2]
        x = x @ p["encoder/continuous_inputs_projection/kernel"] + self.pe[:T]
        for i in range(cfg.num_encoder_layers):
            L = f"encoder/layers_{i}"
            h = rms_norm(x, p[L + "/pre_attention_layer_norm/scale"])
            x = x + self._mha(L + "/attention", h, h)
            h = rms_norm(x, p[L + "/pre_mlp_layer_norm/scale"])
            x = x + self._mlp(L + "/mlp", h)
        return rms_norm(x, p["encoder/encoder_norm/scale"])

    # ---- decoder, teacher-forced (network.py:196-262 with a causal mask)
    def decode_logits(self, encoded: torch.Tensor, dec_in: np.ndarray) -> torch.Tensor:
        p, cfg = self.p, self.cfg
        tok = torch.from_numpy(np.asarray(dec_in, np.int64))
        Lq = tok.shape[1]
        y = p["decoder/token_embedder/embedding"][tok] + self.pe[:Lq]
        causal = torch.full((Lq, Lq), -1e10, dtype=self.dtype).triu(1)[None, None]
        for i in range(cfg.num_decoder_layers):
            L = f"decoder/layers_{i}"
            h = rms_norm(y, p[L + "/pre_self_attention_layer_norm/scale"])
            y = y + self._mha(L + "/self_attention", h, h, causal)
            h = rms_norm(y, p[L + "/pre_cross_attention_layer_norm/scale"])
            y = y + self._mha(L + "/encoder_decoder_attention", h, encoded)
            h = rms_norm(y, p[L + "/pre_mlp_layer_norm/scale"])
            y = y + self._mlp(L + "/mlp", h)
        y = rms_norm(y, p["decoder/decoder_norm/scale"])
        return y @ p["decoder/logits_dense/kernel"]

    # ---- incremental decode with a KV cache (layers.py:246-314 semantics, own layout)
    def _init_cache(self, encoded):
        cfg, p = self.cfg, self.p
        cache = []
        for i in range(cfg.num_decoder_layers):
            L = f"decoder/layers_{i}/encoder_decoder_attention"
            cache.append({
                "ck": self._heads(encoded, p[L + "/key/kernel"]),
                "cv": self._heads(encoded, p[L + "/value/kernel"]),
                "k": [], "v": []})
        return cache

    def _step(self, cache, tok: torch.Tensor, t: int) -> torch.Tensor:
        """One decode step: tokens [B] at position t -> logits [B, V]."""
        p, cfg = self.p, self.cfg
        y = (p["decoder/token_embedder/embedding"][tok] + self.pe[t])[:, None, :]
        for i in range(cfg.num_decoder_layers):
            L = f"decoder/layers_{i}"
            c = cache[i]
            h = rms_norm(y, p[L + "/pre_self_attention_layer_norm/scale"])
            S = L + "/self_attention"
            q = self._heads(h, p[S + "/query/kernel"])
            c["k"].append(self._heads(h, p[S + "/key/kernel"]))
            c["v"].append(self._heads(h, p[S + "/value/kernel"]))
            o = attention(q, torch.cat(c["k"], 1), torch.cat(c["v"], 1))
            y = y + o.reshape(o.shape[0], 1, -1) @ p[S + "/out/kernel"]
            h = rms_norm(y, p[L + "/pre_cross_attention_layer_norm/scale"])
            X = L + "/encoder_decoder_attention"
            q = self._heads(h, p[X + "/query/kernel"])
            o = attention(q, c["ck"], c["cv"])
            y = y + o.reshape(o.shape[0], 1, -1) @ p[X + "/out/kernel"]
            h = rms_norm(y, p[L + "/pre_mlp_layer_norm/scale"])
            y = y + self._mlp(L + "/mlp", h)
        y = rms_norm(y, p["decoder/decoder_norm/scale"])
        return (y @ p["decoder/logits_dense/kernel"])[:, 0, :]

    @torch.no_grad()
    def greedy_decode(self, encoded: torch.Tensor, max_steps: int, eos_id: int = 1,
                      return_logits: bool = False, eos_lengths=None):
        """Greedy: argmax each step (ties -> lowest id), BOS = 0; once a row has
        emitted EOS its later ids are 0 (pad).  Returns int32 [B, max_steps]
        (and the per-step logits if asked).  eos_lengths [B] (tests / bench only): the
        synthetic EOS schedule of SURVEY.md 8(d) -- row b's distribution at step
        eos_lengths[b] - 1 is a point mass on EOS (include/mt3_hip_debug.h states the
        same rule for the engine)."""
        B = encoded.shape[0]
        forced = None if eos_lengths is None else torch.as_tensor(np.asarray(eos_lengths, np.int64))
        cache = self._init_cache(encoded)
        tok = torch.zeros(B, dtype=torch.int64)
        done = torch.zeros(B, dtype=torch.bool)
        ids = torch.zeros(B, max_steps, dtype=torch.int32)
        all_logits = []
        for t in range(max_steps):
            logits = self._step(cache, tok, t)
            if return_logits:
                all_logits.append(logits.clone())
            nxt = torch.argmax(logits, dim=-1)
            if forced is not None:
                nxt = torch.where(t + 1 >= forced, torch.full_like(nxt, eos_id), nxt)
            nxt = torch.where(done, torch.zeros_like(nxt), nxt)
            ids[:, t] = nxt.to(torch.int32)
            done = done | (nxt == eos_id)
            tok = nxt
        if return_logits:
            return ids.numpy(), torch.stack(all_logits, 1)
        return ids.numpy()

    @torch.no_grad()
    def beam1_decode(self, encoded: torch.Tensor, max_steps: int, eos_id: int = 1, alpha: float = 0.6):
        """Emulation of t5x decoding.beam_search with num_decodes=1
        [from memory]: the live hypothesis follows the best NON-EOS token; whenever
        EOS is among the top-2 candidates the prefix+EOS is scored
        logp / ((5+len)/6)^alpha and kept if it beats the best finished one; the row
        stops when the best finished score can no longer be beaten.  If nothing
        finished, the live hypothesis is returned."""
        B = encoded.shape[0]
        out = np.zeros((B, max_steps), np.int32)
        for b in range(B):
            cache = self._init_cache(encoded[b:b + 1])
            tok = torch.zeros(1, dtype=torch.int64)
            live_lp, live, best_fin, best_score = 0.0, [], None, -1e30
            for t in range(max_steps):
                lp = torch.log_softmax(self._step(cache, tok, t)[0].double(), -1)
                top = torch.topk(lp, 2)
                cand = [(float(top.values[j]), int(top.indices[j])) for j in range(2)]
                for v, i in cand:
                    if i == eos_id:
                        score = (live_lp + v) / (((5.0 + t + 1) / 6.0) ** alpha)
                        if score > best_score:
                            best_score, best_fin = score, live + [eos_id]
                v, i = next((v, i) for v, i in cand if i != eos_id)
                live_lp += v
                live.append(i)
                tok = torch.tensor([i])
                bound = live_lp / (((5.0 + max_steps + 1) / 6.0) ** alpha)   # t5x: max_decode_len += 1 (dummy start token)
                if best_fin is not None and best_score > bound:
                    break
            seq = best_fin if best_fin is not None else live
            out[b, : len(seq)] = seq[:max_steps]
        return out
