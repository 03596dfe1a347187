"""Audio ingest: WAV bytes/file -> mono float32 at 16 kHz.

The notebook uses `note_seq.audio_io.wav_data_to_samples_librosa` and `librosa.resample`
(mt3/preprocessors.py:139-144); neither is available here.  This module decodes PCM WAV with the
standard library / scipy and resamples with a polyphase Kaiser filter (`scipy.signal.resample_poly`).
The resampling filter differs from librosa's -- PARITY UNPINNED (inputs already at 16 kHz are
bit-identical).
"""
from __future__ import annotations

import io
from fractions import Fraction

import numpy as np

SAMPLE_RATE = 16000


def wav_data_to_samples(wav_data, sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """bytes or path -> float32 mono in [-1, 1] at `sample_rate`."""
    from scipy.io import wavfile
    src = io.BytesIO(wav_data) if isinstance(wav_data, (bytes, bytearray)) else wav_data
    native_sr, y = wavfile.read(src)
    if y.dtype == np.uint8:
        y = (y.astype(np.float32) - 128.0) / 128.0
    elif np.issubdtype(y.dtype, np.integer):
        y = y.astype(np.float32) / float(np.iinfo(y.dtype).max + 1)
    else:
        y = y.astype(np.float32)
    if y.ndim == 2:
        y = y.mean(axis=1)
    return resample(y, native_sr, sample_rate)


def resample(y: np.ndarray, orig_sr: int, target_sr: int = SAMPLE_RATE) -> np.ndarray:
    if orig_sr == target_sr:
        return np.ascontiguousarray(y, np.float32)
    from scipy.signal import resample_poly
    frac = Fraction(int(target_sr), int(orig_sr))
    return resample_poly(y.astype(np.float64), frac.numerator, frac.denominator).astype(np.float32)


def samples_to_wav_data(samples: np.ndarray, sample_rate: int = SAMPLE_RATE) -> bytes:
    from scipy.io import wavfile
    buf = io.BytesIO()
    wavfile.write(buf, sample_rate, (np.clip(samples, -1, 1) * 32767.0).astype(np.int16))
    return buf.getvalue()
