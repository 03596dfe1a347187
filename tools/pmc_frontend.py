"""Log-mel frontend launches (256 segments) for a rocprofv3 --pmc run: is the kernel bound by VALU issue, by LDS, or
by memory?  (tools/gpurun.sh stage pmcfe)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import spectrograms, synthetic  # noqa: E402

audio = synthetic.synth_audio(256, seed=3)
for _ in range(4):
    x = spectrograms.compute_spectrogram_batch(audio, None)
torch.cuda.synchronize()
print("pmc_frontend done", tuple(x.shape))
