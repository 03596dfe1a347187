"""Round-4 A/B on one box, one process: the two-deep pipeline (frontend + encoder of step i + 1 beside the decode of step i)
against one call at a time, with the pipelined encoder on a quarter / half / all of the compute units -- whole steps
(frontend, encode, decode, ids -> tokens, device -> host copy) in the canonical full-length schedule and under the synthetic
EOS schedule (lengths ~ clipped N(300, 100), early exit + row retirement).  Usage: python tools/ab_r4.py [name-substring ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic, vocabularies  # noqa: E402

B = int(os.environ.get("AB_B", "256"))
N = int(os.environ.get("AB_STEPS", "4"))
K = _lib
VARIANTS = [
    ("one call at a time", False, 0),
    ("two-deep, encoder on every 4th CU", True, 0),
    ("two-deep, encoder on every 2nd CU", True, K.OPT_X_ENCODE_ON_HALF),
    ("two-deep, encoder on all CUs", True, K.OPT_X_ENCODE_UNMASKED),
]
want = sys.argv[1:]
stream = torch.cuda.Stream()
audio = synthetic.synth_audio(B, seed=1000)
lens = np.clip(np.rint(np.random.default_rng(0).normal(300, 100, B)), 1, 1024).astype(np.int32)
vocab = vocabularies.vocabulary_from_codec(vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1)))


def run(eng, n, pipelined, **kw):
    out, in_flight = None, False
    with torch.cuda.stream(stream):
        for _ in range(n):
            eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
            if in_flight:
                out = vocab.decode_tf(eng.decode_wait()).cpu()
            eng.decode(num_steps=1024, wait=False, **kw)
            in_flight = True
            if not pipelined:
                out, in_flight = vocab.decode_tf(eng.decode_wait()).cpu(), False
        if in_flight:
            out = vocab.decode_tf(eng.decode_wait()).cpu()
    torch.cuda.synchronize()
    return out


for dtype in ("float32", "bfloat16"):
    first = None
    for name, pipelined, opt in VARIANTS:
        if want and not any(w in name for w in want):
            continue
        cfg = network.T5Config(dtype=dtype)
        eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B, options=opt)
        eng.load_params(network.init_random_params(cfg, seed=0))
        res = []
        for eos in (False, True):
            eng.debug_set_eos_schedule(lens if eos else None)
            kw = dict(early_exit=True) if eos else {}
            run(eng, 2, pipelined, **kw)
            t0 = time.perf_counter()
            out = run(eng, N, pipelined, **kw)
            res.append(((time.perf_counter() - t0) * 1e3 / N, out))
        eng.debug_set_eos_schedule(None)
        same = ""
        if first is None:
            first = [r[1] for r in res]
        else:
            same = " | tokens equal to the first variant: %s / %s" % (torch.equal(res[0][1], first[0]), torch.equal(res[1][1], first[1]))
        print("%-8s %-36s full-length step %.1f ms (%.1f audio-s/s) | EOS-schedule step %.1f ms (%.1f audio-s/s)%s" % (
            dtype, name, res[0][0], B * 2.048 / res[0][0] * 1e3, res[1][0], B * 2.048 / res[1][0] * 1e3, same), flush=True)
        del eng
