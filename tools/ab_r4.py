"""Round-4 A/B on one box, one process: FIVE and SIX row groups against the product's four (f32, B = 256): whole decodes
at full length and under the synthetic EOS schedule.  (Eight groups were measured earlier in the round: 2.1x slower,
profiles/r4_ab_eight_row_groups.txt.)  Needs the experiment build of libmt3hip.so (mt3_debug_set_groups).
Usage: python tools/ab_r4.py"""
import ctypes
import os
import sys
import resource

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

lib = _lib.load()
lib.mt3_debug_set_groups.argtypes = [ctypes.c_int]
lib.mt3_debug_set_groups.restype = ctypes.c_int
S = 1024


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1), out


def run(dtype, groups, ref):
    lib.mt3_debug_set_groups(groups)
    cfg = network.T5Config(dtype=dtype)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=S, max_batch=256)
    eng.load_params(network.init_random_params(cfg, seed=0))
    eng.encode(spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=1000), None))
    eng.decode(num_steps=4)
    torch.cuda.synchronize()
    best, ids = 1e9, None
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    for _ in range(3):
        ms, ids = timed(lambda: eng.decode(num_steps=S))
        best = min(best, ms)
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = ((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)) / 3
    ngroups = eng.status(_lib.STATUS_LAST_DECODE_GROUPS)
    ids = ids.cpu().numpy()
    rng = np.random.default_rng(7)
    lens = np.clip(np.rint(rng.normal(300, 100, 256)), 1, S).astype(np.int32)
    eng.debug_set_eos_schedule(lens)
    eng.decode(num_steps=S, early_exit=True)
    torch.cuda.synchronize()
    eos = min(timed(lambda: eng.decode(num_steps=S, early_exit=True))[0] for _ in range(3))
    eng.debug_set_eos_schedule(None)
    same = "" if ref is None else " | ids equal to the product schedule's: %s" % bool(np.array_equal(ids, ref))
    print("%s, %d row groups: full decode %.1f ms (host cpu %.2f s) | eos schedule %.1f ms%s"
          % (dtype, ngroups, best, cpu, eos, same), flush=True)
    del eng
    return ids


if __name__ == "__main__":
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ref = run("float32", 0, None)          # the product's choice: 4
    for g in ((5, 6) if quick else (5, 6, 3)):
        run("float32", g, ref)
    if not quick:
        ref = run("bfloat16", 0, None)         # the product's choice: 2
        for g in (3, 4):
            run("bfloat16", g, ref)
