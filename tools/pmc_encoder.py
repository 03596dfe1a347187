"""Encoder pass (B=256) for a rocprofv3 --pmc run: what are the big GEMM / attention waves waiting on?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

cfg = network.T5Config(dtype="bfloat16")
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
eng.load_params(network.init_random_params(cfg, seed=0))
x = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=3), None)
for _ in range(2):
    eng.encode(x)
torch.cuda.synchronize()
print("pmc_encoder done")
