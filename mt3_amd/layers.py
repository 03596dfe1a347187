"""Attention-mask helpers (mirror of the mask builders in mt3/layers.py:627-830).

The inference path never materialises a mask -- the encoder mask is all ones
(network.py:283-289) and the cached decoder's causal mask is the loop bound of the
decode-attention kernel -- but the helpers are part of the reference's layer library
(SURVEY.md 8a row a19), needed by anyone who scores teacher-forced sequences, so
they are provided here in numpy with the reference's signatures and pinned by its
own literals (mt3/layers_test.py:117-283).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def make_attention_mask(query_input, key_input, pairwise_fn: Callable = np.multiply, extra_batch_dims: int = 0,
                        dtype=np.float32):
    """[batch, len_q], [batch, len_kv] -> [batch, 1, len_q, len_kv] (one broadcast head axis)."""
    q, k = np.asarray(query_input), np.asarray(key_input)
    m = pairwise_fn(q[..., :, None], k[..., None, :])[..., None, :, :]
    for _ in range(extra_batch_dims):
        m = m[None]
    return m.astype(dtype)


def make_causal_mask(x, extra_batch_dims: int = 0, dtype=np.float32):
    """Lower-triangular mask of x's trailing length; independent of x's values."""
    x = np.asarray(x)
    idx = np.broadcast_to(np.arange(x.shape[-1], dtype=np.int32), x.shape)
    return make_attention_mask(idx, idx, np.greater_equal, extra_batch_dims=extra_batch_dims, dtype=dtype)


def combine_masks(*masks, dtype=np.float32):
    """Logical AND of the given masks (None entries skipped); None if there is none."""
    ms = [np.asarray(m) for m in masks if m is not None]
    if not ms:
        return None
    assert all(m.ndim == ms[0].ndim for m in ms), "masks must have same rank"
    out = ms[0].astype(bool)
    for m in ms[1:]:
        out = np.logical_and(out, m)
    return out.astype(dtype)


def combine_biases(*biases):
    """Sum of the given additive biases (None entries skipped); None if there is none."""
    bs = [np.asarray(b) for b in biases if b is not None]
    if not bs:
        return None
    assert all(b.ndim == bs[0].ndim for b in bs), "masks must have same rank"
    out = bs[0]
    for b in bs[1:]:
        out = out + b
    return out


def make_decoder_mask(decoder_target_tokens, dtype, decoder_causal_attention: Optional[np.ndarray] = None,
                      decoder_segment_ids: Optional[np.ndarray] = None):
    """causal (or prefix-LM) mask AND padding mask (targets > 0) AND same-segment mask."""
    tgt = np.asarray(decoder_target_tokens)
    parts = []
    causal = make_causal_mask(tgt, dtype=dtype)
    if decoder_causal_attention is not None:
        prefix = make_attention_mask(decoder_causal_attention, decoder_causal_attention, np.logical_and, dtype=dtype)
        parts.append(np.logical_or(causal, prefix).astype(dtype))
    else:
        parts.append(causal)
    parts.append(make_attention_mask(tgt > 0, tgt > 0, dtype=dtype))
    if decoder_segment_ids is not None:
        parts.append(make_attention_mask(decoder_segment_ids, decoder_segment_ids, np.equal, dtype=dtype))
    return combine_masks(*parts, dtype=dtype)
