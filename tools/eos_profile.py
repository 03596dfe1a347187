"""The decode loop under the synthetic EOS schedule (lengths ~ clipped N(300, 100), early exit + row retirement), alone,
for a rocprofv3 --kernel-trace --stats run: which kernels the ragged regime spends its time in.
Usage: python tools/eos_profile.py [float32|bfloat16] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 256
cfg = network.T5Config(dtype=dtype)
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
eng.load_params(network.init_random_params(cfg, seed=0))
lens = np.clip(np.rint(np.random.default_rng(0).normal(300, 100, B)), 1, 1024).astype(np.int32)
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    eng.encode(spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=1000), None))
    eng.debug_set_eos_schedule(lens)
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode(num_steps=1024, early_exit=True)
        torch.cuda.synchronize()
        print("%s eos-schedule decode: %.1f ms, %d steps, %d groups, %d compactions" % (
            dtype, (time.perf_counter() - t0) * 1e3, eng.steps_run, eng.status(_lib.STATUS_LAST_DECODE_GROUPS),
            eng.status(_lib.STATUS_LAST_DECODE_COMPACTIONS)), flush=True)
