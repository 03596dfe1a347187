"""Robustness soak of mt3_engine_transcribe (in-flight batching; round 5): for `--seconds` of wall time, random jobs --
1 .. 2,500 segments through a `--slots`-slot engine (64-segment chunks through the 8-chunk ring), random output
lengths (the synthetic EOS schedule per segment, mixed with rows that emit EOS of their own accord and rows that hit the
step cap), greedy / beam-1, 1 / 2 / 3 / 4 row groups, poll intervals 1 .. 16, graph replay / direct launches, NaN-poisoned
caches every few jobs -- every job's ids compared with plain batch-synchronous calls of the same engine.  Prints one
line per job and a summary; exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--slots", type=int, default=160)
args = ap.parse_args()

L, S = 1024, 96
cfg = network.T5Config(dtype=args.dtype, num_encoder_layers=1, num_decoder_layers=2)
params = network.init_random_params(cfg, seed=3, norm_scale_jitter=0.1)
k = params["decoder/logits_dense/kernel"].copy()
k[:, 1] *= 3.0
params["decoder/logits_dense/kernel"] = k
B = args.slots
eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B)
eng.load_params(params)
pool = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(1024, seed=99), None)
rng = np.random.default_rng(0)
t_end = time.perf_counter() + args.seconds
jobs = segs = 0
while time.perf_counter() < t_end:
    N = int(rng.choice([1, 7, 8, 9, B - 1, B, B + 1, 2 * B + 3, int(rng.integers(1, 2500))]))
    idx = rng.integers(0, pool.shape[0], N)
    lm = pool[torch.from_numpy(idx).cuda()]
    lens = np.clip(np.rint(rng.normal(rng.choice([5, 30, 60]), 25, N)), 1, S + 30).astype(np.int32)
    beam1 = bool(rng.integers(0, 2))
    kw = dict(beam1=beam1, use_graph=bool(rng.integers(0, 4)), debug_poll_steps=int(rng.choice([0, 1, 2, 4, 8, 16])),
              debug_row_groups=int(rng.integers(0, 5)))
    # reference: plain calls of <= B segments, every row every step (a last call of < 8 segments is moved back to cover 8)
    ref = []
    for a in range(0, N, B):
        b = min(a + B, N)
        a0 = max(0, b - 8) if b - a < 8 else a
        eng.encode(lm[a0:b])
        eng.debug_set_eos_schedule(lens[a0:b])
        ref.append(eng.decode(num_steps=S, single_stream=True, beam1=beam1)[a - a0:])
    ref = torch.cat(ref)
    eng.debug_set_eos_schedule(lens)
    if jobs % 5 == 4:
        eng.debug_poison_caches(0xFF, cross=True)
    got = eng.transcribe(lm, num_steps=S, **kw)
    eng.debug_set_eos_schedule(None)
    st = eng.transcribe_stats
    ok = torch.equal(got, ref)
    jobs += 1
    segs += N
    print("job %3d: %4d segments, %s, groups %d, poll %2d, %s: %s  %s" % (
        jobs, N, "beam-1" if beam1 else "greedy", st["groups"], kw["debug_poll_steps"] or 4, "graph" if kw["use_graph"] else "direct",
        "ok" if ok else "MISMATCH rows %s" % (got != ref).any(1).nonzero().flatten().tolist()[:8], st), flush=True)
    if not ok:
        sys.exit(1)
print("soak ok: %d jobs, %d segments, %.0f s; graph fallbacks %d" % (jobs, segs, args.seconds, eng.status(_lib.STATUS_GRAPH_FALLBACKS)))
