#!/usr/bin/env python3
"""Generate tests/golden/network_golden.npz by running the reference's REAL mt3/layers.py and mt3/network.py.

JAX / Flax are not installable in the build container, so the two reference files are imported UNMODIFIED from
/root/reference on top of tests/golden/jax_standin.py, a numpy stand-in for the handful of jax/flax entry points
they use.  What the fixture pins is therefore the reference's own wiring: projection layouts, where the (scale-only)
layer norms and residuals sit, attention logits without 1/sqrt(d), the fixed sinusoidal positions and how decode
mode indexes them, the decode-mode K/V cache (one-hot update, cache_index, causal mask over the cache), the
gated-GELU MLP, the f32 logits head -- while einsum / softmax / tanh-GELU leaves are numpy.

Recorded (small T5 so the fixture stays a few hundred KB): encoder output, teacher-forced decoder logits, and the
logits of the CACHED single-step decode path driven exactly as t5x drives it (cache initialised by one full-length
decode=True call, then one token per call with mutable=['cache']).

Usage (build container only):  python tests/golden/make_network_golden.py
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import jax_standin  # noqa: E402

jax_standin.install()
pkg = types.ModuleType("mt3")          # the real mt3/__init__.py imports the whole training stack
pkg.__path__ = [os.path.join(REF, "mt3")]
sys.modules["mt3"] = pkg
from mt3 import network as ref_network  # noqa: E402  (the reference file, unmodified)

from mt3_amd import network as our_network  # noqa: E402  (only for parameter names / initialisers)

CFG = dict(vocab_size=48, emb_dim=32, num_heads=2, head_dim=8, mlp_dim=64, num_encoder_layers=2, num_decoder_layers=2)
INPUT_DEPTH, T, L, B, SEED = 24, 12, 10, 2, 7


def main():
    ours = our_network.T5Config(dtype="float32", input_depth=INPUT_DEPTH, **CFG)
    params = our_network.init_random_params(ours, seed=SEED, norm_scale_jitter=0.3)
    params = {k: np.asarray(v, np.float32) for k, v in params.items()}
    cfg = ref_network.T5Config(dtype=np.float32, mlp_activations=("gelu", "linear"), dropout_rate=0.1,
                               logits_via_embedding=False, **CFG)
    model = ref_network.Transformer(config=cfg)
    rng = np.random.default_rng(SEED)
    x = rng.standard_normal((B, T, INPUT_DEPTH)).astype(np.float32)
    dec_in = rng.integers(0, CFG["vocab_size"], (B, L)).astype(np.int32)
    dec_in[:, 0] = 0                                                   # BOS
    ones_tgt = np.ones((B, L), np.int32)

    encoded = model.apply({"params": params}, x, enable_dropout=False, method=model.encode)
    logits_tf = model.apply({"params": params}, encoded, x, dec_in, ones_tgt, enable_dropout=False, decode=False,
                            method=model.decode)
    # t5x: the cache is created by ONE decode=True pass over full-length dummy inputs ...
    _, variables = model.apply({"params": params}, np.ones_like(x), np.ones((B, L), np.int32), ones_tgt,
                               enable_dropout=False, decode=True, mutable=["cache"])
    cache = variables["cache"]
    init_index = {k: np.asarray(v).tolist() for k, v in cache.items() if k.endswith("index")}
    # ... and every step feeds one token with the cache mutable
    step_logits = []
    for t in range(L):
        tok = dec_in[:, t:t + 1]
        out, variables = model.apply({"params": params, "cache": cache}, encoded, x, tok, tok, enable_dropout=False,
                                     decode=True, max_decode_length=L, mutable=["cache"], method=model.decode)
        cache = variables["cache"]
        step_logits.append(np.asarray(out)[:, 0])
    logits_cached = np.stack(step_logits, 1)
    print("cache variables:", sorted(cache)[:6], "...", len(cache))
    print("indices after the init pass:", init_index)
    print("teacher-forced vs cached logits max |diff|:", float(np.abs(logits_tf - logits_cached).max()))
    np.savez_compressed(os.path.join(HERE, "network_golden.npz"), x=x, dec_in=dec_in,
                        encoded=np.asarray(encoded, np.float32), logits_teacher_forced=np.asarray(logits_tf, np.float32),
                        logits_cached=logits_cached.astype(np.float32),
                        config=np.array([CFG[k] for k in ("vocab_size", "emb_dim", "num_heads", "head_dim", "mlp_dim",
                                                          "num_encoder_layers", "num_decoder_layers")] +
                                        [INPUT_DEPTH, SEED], np.int64),
                        param_checksum=np.float64(sum(float(np.abs(v).sum()) for v in params.values())))
    print("wrote network_golden.npz")


if __name__ == "__main__":
    main()
