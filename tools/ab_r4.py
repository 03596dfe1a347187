"""Round-4 A/B on one box, one process: the f32 engine's encoder on the bf16 pipes (three bf16 planes per operand, six
products) against the f32 matrix instruction: encoder + cross-K/V ms at B = 256 and B = 64 (HIP events, min of 3 x 5
passes), and whole steps of the headline pipeline.  Usage: python tools/ab_r4.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

audio = synthetic.synth_audio(256, seed=1000)
lm = spectrograms.compute_spectrogram_batch(audio, None)
for name, opt in (("three bf16 planes (default)", 0), ("f32 matrix instruction", _lib.OPT_ENCODER_F32_MFMA)):
    cfg = network.T5Config(dtype="float32")
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256, options=opt)
    eng.load_params(network.init_random_params(cfg, seed=0))
    res = []
    for b in (256, 64):
        eng.encode(lm[:b])
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                eng.encode(lm[:b])
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5)
        res.append(best)
    eng.encode(lm)
    eng.decode(num_steps=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.encode(spectrograms.compute_spectrogram_batch(audio, None))
        ids = eng.decode(num_steps=1024)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 3 * 1e3
    print("f32 encoder, %-28s encoder + cross-K/V %.2f ms at B = 256 (%.0f TF/s f32-equivalent), %.2f ms at B = 64 | "
          "frontend + encode + decode %.1f ms per step" % (name + ":", res[0], 12.214 * 256 / res[0], res[1], step), flush=True)
    del eng
