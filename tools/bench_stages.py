"""Per-stage rates and roofline fractions as SURVEY.md 8(d) defines them (HIP events, inputs resident in HBM):
  frontend  : HBM-bound, 655,360 algorithmic B / segment              -> GB/s / 8,000
  encoder   : MFMA-bound, 10.603 GFLOP / segment (+ 1.611 cross-K/V)   -> TFLOP/s / 2,500 (bf16 dense peak)
  configs[1]: BASELINE's encoder-only case, B = 64 (frontend + encoder), also B = 256
Audio: the tonal synthetic set and a pure uniform(-1, 1) noise run (no empty-energy bins)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

ENC_FLOP, CROSS_FLOP, FE_BYTES = 10.603e9, 1.611e9, 655360


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e-3 / reps


out = {}
cfg = network.T5Config(dtype="bfloat16")
params = network.init_random_params(cfg, seed=0)
for B in (64, 256):
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
    eng.load_params(params)
    for kind in ("tonal", "uniform_noise"):
        audio = synthetic.synth_audio(B, seed=3) if kind == "tonal" else \
            (torch.rand(B, 32768, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)) * 2 - 1)
        t_fe = timed(lambda: spectrograms.compute_spectrogram_batch(audio, None), 20)
        x = spectrograms.compute_spectrogram_batch(audio, None)
        t_enc = timed(lambda: eng.encode(x), 10)                 # encoder + cross-K/V of all decoder layers
        t_both = timed(lambda: eng.encode(spectrograms.compute_spectrogram_batch(audio, None)), 10)
        out["B%d_%s" % (B, kind)] = {
            "frontend_us": t_fe * 1e6, "frontend_GBps": B * FE_BYTES / t_fe / 1e9,
            "frontend_frac_hbm": B * FE_BYTES / t_fe / 8e12,
            "encoder_ms": t_enc * 1e3, "encoder_TFLOPps": B * (ENC_FLOP + CROSS_FLOP) / t_enc / 1e12,
            "encoder_frac_mfma_bf16": B * (ENC_FLOP + CROSS_FLOP) / t_enc / 2.5e15,
            "frontend_plus_encoder_segments_per_s": B / t_both,
            "frontend_plus_encoder_audio_s_per_s": B / t_both * 2.048,
        }
    del eng
print(json.dumps(out, indent=1))
