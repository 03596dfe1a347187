"""Where does a decode-sized GEMM launch spend its time IN SITU?  (round 6.)  Needs the MT3_EXP = 32 build of the library
(tools/ab_r6.py build; `python tools/ab_r6.py phases` puts it in place, runs this file and restores the product): lane 0
of every workgroup of the decode-sized tiles adds its 100 MHz wall-clock spans to per-epilogue accumulators, read here
after (1) the product schedule (B = 256, four row groups side by side), (2) the single-stream debug decode (M = 256, one
kernel at a time), (3) the same with both attention kernels left out (the dense chain alone on the chip) and (4) a
64-slot engine decoding 64 rows on one stream (a row group's launches with nothing beside them)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

lib = _lib.load()
read = lib.mt3_exp_gemm_phases
read.restype = C.c_int
read.argtypes = [C.c_void_p]
NAMES = {1: "cross out-projection + residual (RESID, K = 384, N = 512)", 2: "GEGLU wi (K = 512, N = 2048)",
         7: "self out-projection + residual | cross-q (ResidQ, K = 384, N = 896)",
         8: "fold: wo + residual | next q|k|v|cross-q or logits (ResidS, K = 1024 / 1536, N = 2048)"}


def phases(label):
    torch.cuda.synchronize()
    a = np.zeros((16, 8), dtype=np.uint64)
    assert read(a.ctypes.data) == 0
    print(label, flush=True)
    for epi in sorted(NAMES):
        n = float(a[epi, 0])
        if n == 0:
            continue
        t = a[epi].astype(np.float64) * 0.01 / n       # microseconds per workgroup
        print("    %-86s wgs %9d | entry -> loads requested %5.2f | -> first slice in LDS %5.2f | rest of the K loop %5.2f | "
              "epilogue + stores acknowledged %5.2f | sum %5.2f us" % (NAMES[epi], int(n), t[4], t[1] - t[4], t[2], t[3],
                                                                      t[1] + t[2] + t[3]), flush=True)


STEPS = int(os.environ.get("PH_STEPS", "512"))
cfg = network.T5Config(dtype="float32")
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
eng.load_params(network.init_random_params(cfg, seed=0))
eng.encode(spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=1000), None))
eng.decode(num_steps=64)
phases("(warm-up, discarded)")
eng.decode(num_steps=STEPS)
phases("1. product schedule: B = 256 as %d row groups side by side, %d steps from depth 0" % (eng.status(7), STEPS))
eng.debug_decode(num_steps=STEPS)
phases("2. single-stream debug decode, M = 256, one kernel at a time, %d steps" % STEPS)
eng.debug_decode(num_steps=STEPS, skip_self_attn=True, skip_cross_attn=True)
phases("3. the same without the attention kernels (the dense chain alone)")
del eng
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=64)
eng.load_params(network.init_random_params(cfg, seed=0))
eng.encode(spectrograms.compute_spectrogram_batch(synthetic.synth_audio(64, seed=1000), None))
eng.decode(num_steps=64)
phases("(warm-up, discarded)")
eng.decode(num_steps=STEPS)
phases("4. a 64-slot engine, 64 rows on one stream (%d row groups): a row group's launches with nothing beside them" % eng.status(7))
