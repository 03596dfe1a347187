"""A numpy stand-in for the slice of TensorFlow that mt3/spectral_ops.py and mt3/spectrograms.py use, so that the
reference's REAL frontend files can be imported and executed in a container without TensorFlow
(make_frontend_golden.py).  What such a run pins is the reference's own composition and parameters (frame step
int(2048 * (1 - overlap)), pad_end, fft_length=None, 512 mel bins from 20 Hz to the 7600 Hz default, the <= 0 test
and 1e-5 floor of safe_log, hop-sized framing of `split_audio`).  The tf.signal leaves themselves are restated here
from the TensorFlow documentation (they are third-party code that is not in the reference tree), independently of
oracle/frontend.py: periodic Hann via the cosine formula, frames by explicit slicing, rFFT, and the HTK
triangular filterbank written with per-bin loops.  Test tooling for the build container only.
"""
import math
import sys
import types

import numpy as np


class Shape(tuple):
    def concatenate(self, other):
        return Shape(tuple(self) + tuple(other))

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return Shape(r) if isinstance(i, slice) else r


class Tensor(np.ndarray):
    @property
    def shape(self):
        return Shape(np.ndarray.shape.__get__(self))

    def set_shape(self, s):
        assert tuple(self.shape) == tuple(s), (self.shape, s)

    def numpy(self):
        return np.asarray(self)


def _t(x, dtype=None):
    return np.asarray(x, dtype).view(Tensor)


def frame(signal, frame_length, frame_step, pad_end=False, pad_value=0, axis=-1):
    """tf.signal.frame: frames start every `frame_step`; with pad_end the count is ceil(N / step) and the tail is
    padded with pad_value, without it floor((N - length) / step) + 1."""
    x = np.asarray(signal)
    assert axis in (-1, x.ndim - 1)
    n = x.shape[-1]
    if pad_end:
        count = -(-n // frame_step)
        need = (count - 1) * frame_step + frame_length
        if need > n:
            pad = np.full(x.shape[:-1] + (need - n,), pad_value, x.dtype)
            x = np.concatenate([x, pad], -1)
    else:
        count = max(0, (n - frame_length) // frame_step + 1)
    out = np.empty(x.shape[:-1] + (count, frame_length), x.dtype)
    for i in range(count):
        out[..., i, :] = x[..., i * frame_step: i * frame_step + frame_length]
    return _t(out)


def hann_window(window_length, periodic=True, dtype=np.float32):
    denom = window_length if periodic else window_length - 1
    return _t([0.5 - 0.5 * math.cos(2.0 * math.pi * i / denom) for i in range(window_length)], dtype)


def stft(signals, frame_length, frame_step, fft_length=None, window_fn=hann_window, pad_end=False):
    if fft_length is None:                                  # "the smallest power of 2 enclosing frame_length"
        fft_length = 1 << (int(frame_length) - 1).bit_length()
    frames = frame(signals, frame_length, frame_step, pad_end=pad_end)
    frames = np.asarray(frames) * np.asarray(window_fn(frame_length, dtype=np.asarray(signals).dtype))
    return _t(np.fft.rfft(frames, n=fft_length, axis=-1).astype(np.complex64))


def linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                lower_edge_hertz=125.0, upper_edge_hertz=3800.0, dtype=np.float32):
    """HTK mel scale (1127 ln(1 + f / 700)); the DC bin is dropped from the linear frequencies and restored as a
    zero row; band j rises from edge j to edge j + 1 and falls to edge j + 2, edges equally spaced in mel."""
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    nyquist = sample_rate / 2.0
    lin = [nyquist * k / (num_spectrogram_bins - 1) for k in range(num_spectrogram_bins)]
    edges = [mel(lower_edge_hertz) + (mel(upper_edge_hertz) - mel(lower_edge_hertz)) * i / (num_mel_bins + 1)
             for i in range(num_mel_bins + 2)]
    w = np.zeros((num_spectrogram_bins, num_mel_bins), np.float64)
    for k in range(1, num_spectrogram_bins):
        m = mel(lin[k])
        for j in range(num_mel_bins):
            lo, ce, hi = edges[j], edges[j + 1], edges[j + 2]
            w[k, j] = max(0.0, min((m - lo) / (ce - lo), (hi - m) / (hi - ce)))
    return _t(w, dtype)


def install():
    tf = types.ModuleType("tensorflow")
    tf.Tensor = Tensor
    tf.float32 = np.float32
    tf.cast = lambda x, dtype: _t(x, dtype)
    tf.convert_to_tensor = lambda x, dtype=None: _t(x, dtype)
    tf.squeeze = lambda x, axis=None: _t(np.squeeze(np.asarray(x), axis=axis))
    tf.abs = lambda x: _t(np.abs(np.asarray(x)))
    tf.tensordot = lambda a, b, axes: _t(np.tensordot(np.asarray(a), np.asarray(b), axes))
    tf.where = lambda c, a, b: _t(np.where(np.asarray(c), a, np.asarray(b)))
    tf.reshape = lambda x, shape: _t(np.reshape(np.asarray(x), shape))
    tf.math = types.SimpleNamespace(log=lambda x: _t(np.log(np.asarray(x))))
    tf.signal = types.SimpleNamespace(frame=frame, stft=stft, hann_window=hann_window,
                                      linear_to_mel_weight_matrix=linear_to_mel_weight_matrix)
    compat = types.ModuleType("tensorflow.compat")
    v2 = types.ModuleType("tensorflow.compat.v2")
    v2.__dict__.update({k: v for k, v in tf.__dict__.items() if not k.startswith("__")})
    compat.v2 = v2
    tf.compat = compat
    gin = types.ModuleType("gin")
    gin.register = lambda f=None, **k: f if f is not None else (lambda g: g)
    gin.REQUIRED = object()
    for name, mod in (("tensorflow", tf), ("tensorflow.compat", compat), ("tensorflow.compat.v2", v2), ("gin", gin)):
        sys.modules[name] = mod
