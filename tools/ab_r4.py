"""Round-4 A/B of decode-loop variants on one box, one process: for each named variant (dtype, engine options) build an
engine, encode the same 256 segments, and time (a) the canonical 1024-step greedy decode in the product schedule, (b) the
same decode under the synthetic EOS schedule (lengths ~ clipped N(300, 100), early exit + row retirement).  HIP events on
the launch stream for the device side, getrusage for the host CPU seconds of the whole process (group workers included).
Variants that only change schedules give the first variant's ids bit for bit; the split-K tiles change the summation
order of a GEMM output (f32 round-off), so their free-running ids are compared row by row (identical rows, first
divergence) instead.  Usage: python tools/ab_r4.py [variant-name-substring ...]"""
import os
import resource
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

B = int(os.environ.get("AB_B", "256"))
REPS = int(os.environ.get("AB_REPS", "2"))
K = _lib
VARIANTS = [
    ("f32 default (4 row groups)", "float32", 0),
    ("bf16 default (2 row groups)", "bfloat16", 0),
]
want = sys.argv[1:]
stream = torch.cuda.Stream()
audio = synthetic.synth_audio(B, seed=1000)
lens = np.clip(np.rint(np.random.default_rng(0).normal(300, 100, B)), 1, 1024).astype(np.int32)
first = {}


def timed(fn):
    torch.cuda.synchronize()
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        out = fn()
        e1.record(stream)
    e1.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    return e0.elapsed_time(e1), wall, (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), out


for name, dtype, opt in VARIANTS:
    if want and not any(w in name for w in want):
        continue
    cfg = network.T5Config(dtype=dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B, options=opt)
    eng.load_params(network.init_random_params(cfg, seed=0))
    with torch.cuda.stream(stream):
        eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
        eng.decode(num_steps=2)
    full = min((timed(lambda: eng.decode(num_steps=1024)) for _ in range(REPS)), key=lambda t: t[0])
    groups = eng.status(K.STATUS_LAST_DECODE_GROUPS)
    eng.debug_set_eos_schedule(lens)
    with torch.cuda.stream(stream):
        eng.decode(num_steps=1024, early_exit=True)
    eos = min((timed(lambda: eng.decode(num_steps=1024, early_exit=True)) for _ in range(REPS)), key=lambda t: t[1])
    steps, comp = eng.steps_run, eng.status(K.STATUS_LAST_DECODE_COMPACTIONS)
    eos1 = min((timed(lambda: eng.decode(num_steps=1024, early_exit=True, single_stream=True)) for _ in range(REPS)),
               key=lambda t: t[1])
    eng.debug_set_eos_schedule(None)
    key = (dtype, "full"), (dtype, "eos")
    same = ""
    if key[0] in first:
        a, b = full[3].cpu().numpy(), first[key[0]].cpu().numpy()
        neq = a != b
        rows_same = float((~neq.any(1)).mean())
        fd = np.where(neq.any(1), neq.argmax(1), a.shape[1])
        same = " ids vs first: %.3f of rows identical, median first divergence %s; eos ids equal: %s" % (
            rows_same, int(np.median(fd[neq.any(1)])) if neq.any() else None, torch.equal(eos[3], first[key[1]]))
    else:
        first[key[0]], first[key[1]] = full[3].clone(), eos[3].clone()
    print("%-44s groups %d | full decode %.1f ms (host cpu %.2f s) | eos schedule %.1f ms wall (cpu %.2f s, %d steps, %d "
          "compactions) single stream %.1f ms |%s" % (name, groups, full[0], full[2], eos[1], eos[2], steps, comp, eos1[1], same),
          flush=True)
    del eng
