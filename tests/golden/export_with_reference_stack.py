#!/usr/bin/env python3
"""Pin the parts of the oracle that are restated FROM MEMORY of third-party code (tensorflow, jax/flax
and t5x are not installable in the build container) by running the REAL stack once, elsewhere, and dropping the
results into tests/golden/external/.  tests/test_external_fixtures.py picks the files up when they exist and
otherwise reports each check as xfail("fixture absent").

Run on any machine that has the reference's own environment (magenta/mt3 setup.py: tensorflow, jax, flax, t5x,
seqio; a checkout of magenta/mt3; optionally one checkpoint directory from gs://mt3/checkpoints):

    python tests/golden/export_with_reference_stack.py --mt3 /path/to/mt3_checkout \
           [--checkpoint /path/to/checkpoints/mt3] [--only frontend,beam,checkpoint]

Writes (all small):
  external/tf_frontend.npz      (a) the reference's spectrograms.compute_spectrogram (tf.signal.stft / hann_window /
                                linear_to_mel_weight_matrix / log) on the committed synthetic segments
                                (oracle.frontend.synth_audio(4, seed=0)), + the f32 mel matrix TF builds
  external/t5x_decode.npz       (b) the committed tiny model (tests/golden/tiny_model.py) through the real
                                mt3.network.Transformer on flax: encoder output, teacher-forced logits, and
                                t5x.decoding.beam_search (num_decodes=1, alpha=0.6) + temperature_sample(topk=1) ids
  external/t5x_checkpoint/      (c) from a real checkpoint: the msgpack index file verbatim, every parameter's
                                `.zarray`, and the complete data of two small arrays (chunk files verbatim +
                                their values as .npy read through tensorstore/t5x)
Nothing in the product or the -m gpu tests depends on these files; they only let the parity claims marked
"unpinned" in DESIGN.md section 4 be closed by whoever has the stack.
"""
import argparse
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "external")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def export_frontend(mt3_path):
    sys.path.insert(0, mt3_path)
    import tensorflow as tf
    from mt3 import spectrograms
    from oracle import frontend as OF
    cfg = spectrograms.SpectrogramConfig()
    audio = OF.synth_audio(4, seed=0)                         # [4, 32768] f32, deterministic numpy
    audio[3, 20000:] = 0.0                                    # a segment with a silent tail (log floor path)
    logmel = np.stack([np.asarray(spectrograms.compute_spectrogram(a, cfg)) for a in audio])
    mel = tf.signal.linear_to_mel_weight_matrix(spectrograms.input_depth(cfg), spectrograms.FFT_SIZE // 2 + 1,
                                                cfg.sample_rate, spectrograms.MEL_LO_HZ, 7600.0).numpy()
    hann = tf.signal.hann_window(spectrograms.FFT_SIZE).numpy()
    np.savez_compressed(os.path.join(OUT, "tf_frontend.npz"), audio=audio, logmel=logmel.astype(np.float32),
                        mel_matrix=mel.astype(np.float32), hann=hann.astype(np.float32),
                        tf_version=np.array(tf.__version__))
    print("frontend:", logmel.shape, "tf", tf.__version__)


def export_beam(mt3_path):
    sys.path.insert(0, mt3_path)
    import jax
    import jax.numpy as jnp
    from mt3 import network
    from t5x import decoding
    import tiny_model as TM
    cfg = network.T5Config(dtype=jnp.float32, mlp_activations=("gelu", "linear"), dropout_rate=0.1,
                           logits_via_embedding=False, **TM.CFG)
    model = network.Transformer(config=cfg)
    params = jax.tree_util.tree_map(jnp.asarray, TM.nested(TM.params()))
    x, forced = TM.inputs()
    B, L = forced.shape
    encoded = model.apply({"params": params}, x, enable_dropout=False, method=model.encode)
    dec_in = np.concatenate([np.zeros((B, 1), np.int32), forced[:, :-1]], 1)
    logits_tf = model.apply({"params": params}, encoded, x, dec_in, np.ones((B, L), np.int32), enable_dropout=False,
                            decode=False, method=model.decode)
    # the cached one-token path exactly as t5x's predict_batch_with_aux drives it (models.py:121-152 + t5x)
    _, init = model.apply({"params": params}, jnp.ones_like(x), jnp.ones((B, L), jnp.int32),
                          jnp.ones((B, L), jnp.int32), decode=True, enable_dropout=False, mutable=["cache"])
    cache = init["cache"]

    def tokens_to_logits(state):
        flat_logits, new_vars = model.apply({"params": params, "cache": state.cache}, encoded, x, state.cur_token,
                                            state.cur_token, enable_dropout=False, decode=True, max_decode_length=L,
                                            mutable=["cache"], method=model.decode)
        return jnp.squeeze(flat_logits, axis=1), new_vars["cache"]

    prompt = jnp.zeros((B, L), jnp.int32)
    beam, beam_scores = decoding.beam_search(inputs=prompt, cache=cache, tokens_to_logits=tokens_to_logits, eos_id=1,
                                             num_decodes=1, alpha=0.6, max_decode_len=L)
    greedy, greedy_lp = decoding.temperature_sample(inputs=prompt, cache=cache, tokens_to_logits=tokens_to_logits,
                                                    eos_id=1, num_decodes=1, topk=1, temperature=0.0,
                                                    decode_rng=jax.random.PRNGKey(0))
    np.savez_compressed(os.path.join(OUT, "t5x_decode.npz"), encoded=np.asarray(encoded, np.float32),
                        logits_teacher_forced=np.asarray(logits_tf, np.float32),
                        beam_ids=np.asarray(beam[:, -1], np.int32), beam_scores=np.asarray(beam_scores[:, -1]),
                        greedy_ids=np.asarray(greedy[:, -1] if greedy.ndim == 3 else greedy, np.int32),
                        jax_version=np.array(jax.__version__))
    print("decode: beam ids", np.asarray(beam).shape)


def export_checkpoint(ckpt_dir):
    from t5x import checkpoints as t5x_ckpt                    # noqa: F401  (proves the t5x side can read it)
    dst = os.path.join(OUT, "t5x_checkpoint")
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    shutil.copy(os.path.join(ckpt_dir, "checkpoint"), os.path.join(dst, "checkpoint"))
    wanted = ("target.encoder.encoder_norm.scale", "target.decoder.layers_0.pre_mlp_layer_norm.scale")
    listing = {}
    for name in sorted(os.listdir(ckpt_dir)):
        p = os.path.join(ckpt_dir, name)
        if not (os.path.isdir(p) and os.path.exists(os.path.join(p, ".zarray"))):
            continue
        with open(os.path.join(p, ".zarray")) as f:
            listing[name] = json.load(f)
        os.makedirs(os.path.join(dst, name))
        shutil.copy(os.path.join(p, ".zarray"), os.path.join(dst, name, ".zarray"))
        if name in wanted:
            for chunk in os.listdir(p):
                shutil.copy(os.path.join(p, chunk), os.path.join(dst, name, chunk))
            import tensorstore as ts
            arr = ts.open({"driver": "zarr", "kvstore": {"driver": "file", "path": p}}).result().read().result()
            np.save(os.path.join(dst, name + ".values.npy"), np.asarray(arr))
    with open(os.path.join(dst, "listing.json"), "w") as f:
        json.dump(listing, f, indent=1)
    print("checkpoint:", len(listing), "arrays listed")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mt3", required=True, help="path of a magenta/mt3 checkout (the directory that holds mt3/)")
    ap.add_argument("--checkpoint", default=None, help="a t5x checkpoint directory, e.g. checkpoints/mt3/")
    ap.add_argument("--only", default="frontend,beam,checkpoint")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    which = set(args.only.split(","))
    if "frontend" in which:
        export_frontend(args.mt3)
    if "beam" in which:
        export_beam(args.mt3)
    if "checkpoint" in which and args.checkpoint:
        export_checkpoint(args.checkpoint)


if __name__ == "__main__":
    main()
