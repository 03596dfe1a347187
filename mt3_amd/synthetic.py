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
                      num_velocity_bins: int = 1, velocity: float = 1.0):
    """Random-init weights that DECODE NOTES (smoke / end-to-end tests only; there is no checkpoint offline): the logits
    columns of the tokens a note needs are scaled -- `tie` (ends the tie section a segment starts in,
    mt3/note_sequences.py:313-408), the 128 pitches, the first 200 time shifts, EOS -- so that a greedy / beam-1 decode of
    random weights walks through valid note events instead of the flat soup random logits give (58 notes per 262,144
    tokens).  Token ids follow vocabularies.build_codec (mt3/vocabularies.py:119-140): id = 3 + event index; shift
    0..1000 | pitch | velocity | tie | program | drum.  `velocity` scales the velocity tokens (the `ismir2021` preset has 127 bins and no
    tie section: its note-offs are velocity-0 tokens).  Returns a new dict."""
    out = dict(params)
    k = params["decoder/logits_dense/kernel"].copy()
    first_pitch = 3 + 1001
    first_vel = first_pitch + 128
    tie_id = first_vel + num_velocity_bins + 1
    k[:, 1] *= eos
    k[:, 3 + 1: 3 + 201] *= shift
    k[:, first_pitch: first_pitch + 128] *= pitch
    k[:, tie_id] *= tie
    k[:, first_vel: first_vel + num_velocity_bins + 1] *= velocity       # velocity 0 (= note-off) .. num_velocity_bins
    out["decoder/logits_dense/kernel"] = k
    return out


# ------------------------------------------------------------------------------------------------------------------
# Synthetic MUSIC: note lists with known onsets / offsets / pitches and the audio rendered from them.  Two users:
# tools/train_synthetic.py (training pairs for the checkpoint under tests/golden/) and the note-level evaluations of
# bench.py / tests (the file whose ground truth is known).  Not on the product path.
def random_music(seconds: float, seed: int = 0, notes_per_second: float = 6.0, min_pitch: int = 36, max_pitch: int = 96,
                 max_polyphony: int = 6):
    """A random piece as a NoteSequence (program 0, velocity 100): onsets uniform over the piece, durations log-uniform
    0.12 .. 1.6 s, pitches uniform; a note that would overlap (or start within 60 ms of the end of) a sounding note of the
    same pitch, or exceed `max_polyphony` sounding notes, is dropped.  Deterministic in `seed` (numpy only)."""
    import numpy as np
    from . import note_sequences as NS
    rng = np.random.default_rng(seed)
    n = int(rng.poisson(notes_per_second * seconds))
    on = np.sort(rng.uniform(0.0, max(0.05, seconds - 0.15), n))
    dur = np.exp(rng.uniform(np.log(0.12), np.log(1.6), n))
    pitch = rng.integers(min_pitch, max_pitch + 1, n)
    ns = NS.NoteSequence()
    sounding = []                                        # (end, pitch) of notes that may still sound
    for a, d, p in zip(on, dur, pitch):
        b = min(a + d, seconds - 0.02)
        if b - a < 0.1:
            continue
        sounding = [(e, q) for e, q in sounding if e + 0.06 > a]
        if len(sounding) >= max_polyphony or any(q == p for _, q in sounding):
            continue
        sounding.append((b, int(p)))
        ns.notes.append(NS.Note(float(a), float(b), int(p), 100, 0, False, 0))
    ns.total_time = max([n_.end_time for n_ in ns.notes], default=0.0)
    return ns


def render_notes(onsets, offsets, pitches, amps, file_index, n_files: int, n_samples: int, seed: int = 0,
                 sample_rate: int = 16000, device: str = "cuda", noise: float = 0.002):
    """Audio of `n_files` files of `n_samples` samples from flat note arrays (note i belongs to file `file_index[i]`):
    each note is a harmonic tone (f0 of its MIDI pitch, partials k = 1..6 at 1/k below 7.6 kHz, random phases) under a
    piano-like envelope -- 4 ms attack, exp(-t / 0.7 s) decay, 40 ms release after the offset -- rendered over its own span
    and scatter-added into its file; a little white noise; every file peak-normalised to 0.9 (mt3/mixing.py:71-75).
    Returns f32 [n_files, n_samples] on `device`."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    on = torch.as_tensor(onsets, dtype=torch.float64, device=device)
    off = torch.as_tensor(offsets, dtype=torch.float64, device=device)
    out = torch.zeros(n_files * n_samples, device=device)
    if on.numel():
        p = torch.as_tensor(pitches, dtype=torch.float32, device=device)
        amp = torch.as_tensor(amps, dtype=torch.float32, device=device)
        fi = torch.as_tensor(file_index, dtype=torch.long, device=device)
        release = 0.04
        span = int((float((off - on).max()) + release) * sample_rate) + 2
        f0 = 440.0 * torch.exp2((p - 69.0) / 12.0)
        s0 = torch.floor(on * sample_rate).long()                                  # first sample of the note's span
        for a in range(0, on.numel(), 512):                                        # 512 notes x span samples at a time
            sl = slice(a, a + 512)
            idx = s0[sl, None] + torch.arange(span, device=device)[None]           # [n, span] sample index in the file
            t = (idx.double() / sample_rate - on[sl, None]).float()                # time since the onset
            d = (off[sl] - on[sl]).float()[:, None]
            env = (t / 0.004).clamp(0.0, 1.0) * torch.exp(-t.clamp_min(0.0) / 0.7) * \
                (1.0 - (t - d) / release).clamp(0.0, 1.0) * (t >= 0)
            x = torch.zeros_like(t)
            ph = torch.rand((idx.shape[0], 6), device=device, generator=g) * (2 * math.pi)
            for k in range(1, 7):
                ok = (f0[sl] * k < 7600.0).float()[:, None]
                x += ok * torch.sin(2 * math.pi * (f0[sl, None] * k) * t + ph[:, k - 1: k]) / k
            x *= env * amp[sl, None]
            valid = (idx < n_samples) & (idx >= 0)
            flat = (fi[sl, None] * n_samples + idx.clamp(0, n_samples - 1))[valid]
            out.index_add_(0, flat, x[valid])
    out = out.reshape(n_files, n_samples)
    out += noise * torch.randn(out.shape, device=device, generator=g)
    out *= 0.9 / out.abs().amax(1, keepdim=True).clamp_min(1e-9)
    return out


def synth_music(seconds: float, seed: int = 0, device: str = "cuda", **music):
    """(truth NoteSequence, 16 kHz samples as float32 numpy [n]) of one random piece: `random_music` rendered by
    `render_notes`.  Amplitudes 0.3 .. 1.0 per note (not encoded: one velocity bin in the `mt3` vocabulary)."""
    import numpy as np
    ns = random_music(seconds, seed=seed, **music)
    rng = np.random.default_rng(seed + 1)
    amps = rng.uniform(0.3, 1.0, len(ns.notes))
    n = int(round(seconds * 16000))
    wav = render_notes([n_.start_time for n_ in ns.notes], [n_.end_time for n_ in ns.notes],
                       [n_.pitch for n_ in ns.notes], amps, np.zeros(len(ns.notes), np.int64), 1, n, seed=seed,
                       device=device)
    return ns, wav.reshape(-1).cpu().numpy()


_STUB_BASE = {}


def stub_token_rows(first: int, count: int, length: int = 1024, file_segments: int = 256, piece_segments: int = 32,
                    notes_per_second: float = 38.0):
    """VALID `decode_tf`-form token rows (int32 [count, length], -1 from EOS on) of global segments [first, first + count)
    WITHOUT a model: what stands in for frontend + engine in the multi-rank plumbing tests and `bench.py --dry-run`.
    One dense random piece of `piece_segments` segments (`random_music`: ~300 tokens per segment, the output length of
    SURVEY.md 8(d)'s EOS schedule; it starts and ends in silence, so it can follow itself) is tokenised by the encode side of
    the codec (run_length_encoding.segment_targets: tie sections, run-length shifts, redundant state changes removed --
    mt3/tasks.py:142-178's chain).  Segment g of the corpus is row (g % file_segments) % piece_segments of it with every pitch
    token transposed by (g // file_segments) % 12 - 6 semitones: files differ, every file starts with an empty tie section,
    and a row depends on g alone (not on how the corpus is sharded)."""
    import numpy as np
    from . import note_sequences as NS, run_length_encoding as RLE, vocabularies
    key = (length, piece_segments, notes_per_second)
    if key not in _STUB_BASE:
        codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
        ns = random_music(piece_segments * 2.048, seed=20260930, notes_per_second=notes_per_second, min_pitch=30,
                          max_pitch=100, max_polyphony=40)
        times, values = NS.note_sequence_to_onsets_and_offsets_and_programs(ns)
        n_frames = piece_segments * 256
        ev, si, ei, se, sidx = RLE.encode_and_index_events(NS.NoteEncodingState(), times, values, NS.note_event_data_to_events,
                                                           codec, np.arange(n_frames) / 125.0, NS.note_encoding_state_to_events)
        base = np.full((piece_segments, length), -1, np.int32)
        for s in range(piece_segments):
            t = RLE.segment_targets(ev, si, ei, se, sidx, s * 256, (s + 1) * 256, codec, True)
            t = RLE.remove_redundant_state_changes(t, codec, ("velocity", "program"))[: length - 1]
            base[s, : len(t)] = t
        lo, hi = codec.event_type_range("pitch")
        _STUB_BASE[key] = (base, (base >= lo) & (base <= hi))
    base, is_pitch = _STUB_BASE[key]
    g = np.arange(first, first + count)
    row = (g % file_segments) % piece_segments
    rows = base[row].copy()
    rows += is_pitch[row] * ((g // file_segments) % 12 - 6).astype(np.int32)[:, None]
    return rows
