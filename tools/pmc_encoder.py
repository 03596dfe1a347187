"""Encoder passes (B=256) for a rocprofv3 --pmc run: what are the big GEMM / attention waves waiting on?
Three engines in one process: the bf16 encoder (gemm_glds_kernel), the MXFP8 encoder (gemm_mx8_kernel) and the f32
encoder (gemm_x6_kernel: three bf16 planes per operand; enc_attn_kernel<float>); their GEMM kernels have different names."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import network, spectrograms, synthetic  # noqa: E402

x = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=3), None)
for dense in ("", "fp8_e4m3"):
    cfg = dataclasses.replace(network.T5Config(dtype="bfloat16"), dense_dtype=dense)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
    eng.load_params(network.init_random_params(cfg, seed=0))
    for _ in range(2):
        eng.encode(x)
    torch.cuda.synchronize()
    del eng
cfg = network.T5Config(dtype="float32")
eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
eng.load_params(network.init_random_params(cfg, seed=0))
for _ in range(2):
    eng.encode(x)
torch.cuda.synchronize()
del eng
print("pmc_encoder done")
