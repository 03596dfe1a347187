"""Robustness soak of the row-group decode path: 60 back-to-back decodes (greedy / beam-1 / early exit alternating)
on one engine, ids compared with the single-stream schedule every time; then InferenceModel end to end with
batch_size 256 (early exit, beam-1) on 70 s of synthetic audio."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, inference, network, spectrograms, synthetic  # noqa: E402

cfg = network.T5Config(dtype="bfloat16")
params = network.init_random_params(cfg, seed=0)
k = params["decoder/logits_dense/kernel"].copy()
k[:, 1] *= 2.5
params["decoder/logits_dense/kernel"] = k
B = 256
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
eng.load_params(params)
t0 = time.perf_counter()
for it in range(60):
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=it), None)
    eng.encode(lm)
    kw = [dict(), dict(beam1=True), dict(early_exit=True), dict(beam1=True, early_exit=True)][it % 4]
    n = 1024 if it % 10 == 0 else 160
    a = eng.decode(num_steps=n, **kw)
    assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 2
    b = eng.decode(num_steps=n, single_stream=True, **kw)
    assert torch.equal(a, b), (it, kw)
torch.cuda.synchronize()
print("60 partitioned decodes == single-stream decodes, %.1f s; fallbacks %d / %d" % (
    time.perf_counter() - t0, eng.status(_lib.STATUS_PARTITION_FALLBACKS), eng.status(_lib.STATUS_GRAPH_FALLBACKS)), flush=True)
del eng
m = inference.InferenceModel("random:0", "mt3", dtype="bfloat16", batch_size=256, early_exit=True)
audio = synthetic.synth_audio(300, seed=5).reshape(-1).cpu().numpy()[: 300 * 32768 - 1234]
t0 = time.perf_counter()
ns = m(audio)
print("InferenceModel(batch_size=256) on %.0f s of audio: %d notes, %.1f s wall" % (len(audio) / 16000, len(ns.notes),
                                                                                  time.perf_counter() - t0), flush=True)
