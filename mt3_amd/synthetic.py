"""Synthetic 16 kHz audio for benchmarks and smoke tests (no datasets on the box).

Recipe: per 2.048 s segment a sum of 1-6 harmonic tones (f0 log-uniform
55..1760 Hz, 8 partials at 1/k amplitude, random onset/offset inside the segment) plus one
20 ms white-noise burst per 0.25 s, peak-normalised to 0.9 (the inf-norm normalisation of
mt3/mixing.py:71-75).  Generated on the GPU with torch so that batch-256 inputs take
milliseconds, and returned as a CUDA tensor: bench inputs are HBM-resident by construction.
"""
from __future__ import annotations

import math


def synth_audio(n_segments: int, seed: int = 0, seg_samples: int = 32768, sample_rate: int = 16000,
                device: str = "cuda"):
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    S, N = n_segments, seg_samples
    t = torch.arange(N, device=device, dtype=torch.float32) / sample_rate           # [N]
    dur = N / sample_rate
    n_tones = torch.randint(1, 7, (S,), device=device, generator=g)
    x = torch.zeros(S, N, device=device)
    for tone in range(6):
        on = (n_tones > tone).float()[:, None]
        f0 = torch.exp(torch.rand(S, device=device, generator=g) * (math.log(1760.0) - math.log(55.0))
                       + math.log(55.0))[:, None]
        ab = torch.rand(S, 2, device=device, generator=g) * dur
        a, b = ab.min(1).values[:, None], ab.max(1).values[:, None]
        env = ((t[None] >= a) & (t[None] <= b)).float() * on
        ph = torch.rand(S, 1, device=device, generator=g) * 2 * math.pi
        for k in range(1, 9):
            ok = (f0 * k < sample_rate / 2).float()
            x += env * ok * torch.sin(2 * math.pi * f0 * k * t[None] + ph * k) / k
    burst = int(0.020 * sample_rate)
    nb = int(dur / 0.25)
    noise = torch.randn(S, nb, burst, device=device, generator=g) * 0.5
    starts = ((torch.arange(nb, device=device)[None] * 0.25 + torch.rand(S, nb, device=device, generator=g) * 0.2)
              * sample_rate).long().clamp_(0, N - burst)
    idx = starts[:, :, None] + torch.arange(burst, device=device)[None, None]
    x.scatter_add_(1, idx.reshape(S, -1), noise.reshape(S, -1))
    x *= 0.9 / x.abs().amax(1, keepdim=True).clamp_min(1e-9)
    return x
