"""Frontend kernel alone: segments/s and achieved fraction of the HBM roofline (655,360 algorithmic B/segment)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mt3_amd import spectrograms, synthetic
for S in (256, 2048):
    audio = synthetic.synth_audio(S, seed=3)
    for _ in range(3):
        out = spectrograms.compute_spectrogram_batch(audio, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        out = spectrograms.compute_spectrogram_batch(audio, None)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) * 1e3 / 10
    gbs = S * 655360 / (us * 1e-6) / 1e9
    print(f"frontend S={S}: {us:.1f} us per launch, {S / (us * 1e-6):.0f} segments/s, {gbs:.0f} GB/s algorithmic = {gbs / 8000:.3f} of HBM peak")
