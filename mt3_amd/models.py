"""Feature conversion of the model wrapper (mirror of mt3/models.py:24-118,
`ContinuousInputsEncDecFeatureConverter` with pack=False): every example's continuous
`inputs` [n, depth] are trimmed / zero-padded to `task_feature_lengths['inputs']` rows and the
decoder gets all-zero token rows of `task_feature_lengths['targets']` (at inference the targets are
the dummy empty array of preprocessors.add_dummy_targets).  The zero rows are added AFTER the
log-mel (0.0, not log(1e-5))."""
from __future__ import annotations

from typing import Dict, Mapping, Sequence

import numpy as np


def convert_features(examples: Sequence[Mapping[str, np.ndarray]], task_feature_lengths: Mapping[str, int]
                     ) -> Dict[str, np.ndarray]:
    T, L = task_feature_lengths["inputs"], task_feature_lengths["targets"]
    depth = examples[0]["inputs"].shape[-1] if examples else 0
    enc = np.zeros((len(examples), T, depth), np.float32)
    tgt = np.zeros((len(examples), L), np.int32)
    for i, ex in enumerate(examples):
        x = np.asarray(ex["inputs"], np.float32)[:T]
        enc[i, : x.shape[0]] = x
        t = np.asarray(ex.get("targets", np.zeros((0,), np.int32)), np.int32)[:L]
        tgt[i, : t.shape[0]] = t
    dec_in = np.zeros_like(tgt)
    dec_in[:, 1:] = tgt[:, :-1]                            # seqio autoregressive_inputs: shift right, BOS = 0
    return {"encoder_input_tokens": enc, "decoder_target_tokens": tgt, "decoder_input_tokens": dec_in,
            "decoder_loss_weights": (tgt > 0).astype(np.int32)}
