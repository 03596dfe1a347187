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
                device: str = "cuda", tones: int = 0):
    """tones = 0: 1-6 tones per segment (the MT3 recipe); tones = n: exactly n in every segment."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    S, N = n_segments, seg_samples
    t = torch.arange(N, device=device, dtype=torch.float32) / sample_rate           # [N]
    dur = N / sample_rate
    n_tones = torch.randint(1, 7, (S,), device=device, generator=g)
    if tones:
        n_tones = torch.full_like(n_tones, tones)
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


def synth_slakh_shaped(n_segments: int, seed: int = 0, seg_frames: int = 256, hop: int = 128, max_file_segments: int = 8):
    """"Slakh-shaped" synthetic audio (BASELINE configs[4]): mixed-instrument tracks = SIX tones in every
    segment plus the noise bursts, cut into FILES of any length >= 1 segment (the reference resamples Slakh `mix` files to
    16 kHz and splits them into input-length segments, mt3/preprocessors.py:500-503, NB:331): the segments of a file are
    consecutive, the last one is ragged (1 .. seg_frames frames of audio, zeros after).  Returns (CUDA f32
    [n_segments, seg_frames * hop], numpy int32 true frame counts [n_segments], list of (first, count) per file)."""
    import numpy as np
    audio = synth_audio(n_segments, seed=seed, seg_samples=seg_frames * hop, tones=6)
    rng = np.random.default_rng(seed)
    n_frames = np.full(n_segments, seg_frames, np.int32)
    files, first = [], 0
    while first < n_segments:
        count = int(min(rng.integers(1, max_file_segments + 1), n_segments - first))
        last = first + count - 1
        n_frames[last] = int(rng.integers(1, seg_frames + 1))
        audio[last, int(n_frames[last]) * hop:] = 0.0
        files.append((first, count))
        first += count
    return audio, n_frames, files


def boost_note_events(params, tie: float = 6.0, pitch: float = 2.5, shift: float = 2.0, eos: float = 2.5,
                      num_velocity_bins: int = 1):
    """Random-init weights that DECODE NOTES (smoke / end-to-end tests only; there is no checkpoint offline): the logits
    columns of the tokens a note needs are scaled -- `tie` (ends the tie section a segment starts in,
    mt3/note_sequences.py:313-408), the 128 pitches, the first 200 time shifts, EOS -- so that a greedy / beam-1 decode of
    random weights walks through valid note events instead of the flat soup random logits give (58 notes per 262,144
    tokens).  Token ids follow vocabularies.build_codec (mt3/vocabularies.py:119-140): id = 3 + event index; shift
    0..1000 | pitch | velocity | tie | program | drum.  Returns a new dict."""
    out = dict(params)
    k = params["decoder/logits_dense/kernel"].copy()
    first_pitch = 3 + 1001
    first_vel = first_pitch + 128
    tie_id = first_vel + num_velocity_bins + 1
    k[:, 1] *= eos
    k[:, 3 + 1: 3 + 201] *= shift
    k[:, first_pitch: first_pitch + 128] *= pitch
    k[:, tie_id] *= tie
    out["decoder/logits_dense/kernel"] = k
    return out
