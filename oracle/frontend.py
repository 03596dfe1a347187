"""Oracle (CPU, numpy) for the log-mel frontend.  TEST INFRASTRUCTURE (oracle/__init__.py).

Restates   mt3/spectrograms.py:23-82   (constants, compute_spectrogram, framing)
           mt3/spectral_ops.py:29-88   (safe_log, stft, compute_mag, compute_mel, compute_logmel)
whose arithmetic lives in TensorFlow (`tf.signal.frame/stft/hann_window/
linear_to_mel_weight_matrix`) -- not installable here, so this file restates the
published definitions of those ops [from memory] and is **PARITY UNPINNED**
against TF itself (the reference has no frontend test either).  What is pinned:
the reference's own composition and parameters -- its real spectrograms.py /
spectral_ops.py run unmodified on a numpy stand-in for TensorFlow and agree with
this file to 2e-6 x peak (tests/golden/make_frontend_golden.py,
tests/test_oracle_frontend_golden.py); torch.stft / scipy / a from-the-docs HTK
filterbank agree with the leaves (tests/test_oracle_cross_checks.py); internal
consistency (f32 vs f64 noise floor), linearity, and the mel-matrix structure quoted
in SURVEY.md A.2 (1934 nnz, <=2 per row, 2 empty columns).

    frames   : frame i = x[i*hop : i*hop + fft] zero-padded past the end
               (tf.signal.stft(frame_length=2048, frame_step=128, pad_end=True);
               left-aligned, NOT centred), ceil(N/hop) frames
    window   : periodic Hann  0.5 - 0.5 cos(2 pi k / 2048)
    spectrum : |rfft_2048|                        -> [frames, 1025]
    mel      : HTK mel triangles, 512 bins, 20..7600 Hz, DC row zero, no area norm
    log      : log(where(x <= 0, 1e-5, x))
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000          # spectrograms.py:23
HOP_WIDTH = 128              # spectrograms.py:24
NUM_MEL_BINS = 512           # spectrograms.py:25
FFT_SIZE = 2048              # spectrograms.py:28
MEL_LO_HZ = 20.0             # spectrograms.py:29
MEL_HI_HZ = 7600.0           # spectral_ops.py:79 (compute_logmel default hi_hz)
LOG_EPS = 1e-5               # spectral_ops.py:29


def hertz_to_mel(f):
    return 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)


def mel_weight_matrix(num_mel_bins=NUM_MEL_BINS, num_spectrogram_bins=FFT_SIZE // 2 + 1,
                      sample_rate=SAMPLE_RATE, lo_hz=MEL_LO_HZ, hi_hz=MEL_HI_HZ) -> np.ndarray:
    """tf.signal.linear_to_mel_weight_matrix [third-party, from memory], float64."""
    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]        # DC bin dropped
    spec_mel = hertz_to_mel(lin)[:, None]
    edges = np.linspace(hertz_to_mel(lo_hz), hertz_to_mel(hi_hz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[1, 0], [0, 0]])                                # re-add zero DC row


def hann_periodic(n=FFT_SIZE) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def frame_signal(x: np.ndarray, frame_length=FFT_SIZE, frame_step=HOP_WIDTH) -> np.ndarray:
    """tf.signal.frame(pad_end=True): ceil(N/step) frames, zero padded."""
    n = len(x)
    num = -(-n // frame_step)
    padded = np.zeros(((num - 1) * frame_step + frame_length,) if num else (0,), x.dtype)
    padded[:n] = x
    idx = np.arange(frame_length)[None, :] + frame_step * np.arange(num)[:, None]
    return padded[idx] if num else np.zeros((0, frame_length), x.dtype)


def compute_logmel(samples: np.ndarray, dtype=np.float64, tables: str = "float64") -> np.ndarray:
    """spectral_ops.compute_logmel on ONE segment's flattened samples -> [frames, 512].

    dtype=float64: the "true" value; dtype=float32: every stage rounded to f32 the
    way a float32 TF graph would (FFT itself evaluated in f64 then rounded, i.e. a
    best-case f32 FFT).  tables: how the two constant tables were BUILT -- "float64"
    (the mathematical definition, rounded once) or "tf32" (float32 in TensorFlow's op
    order, hann_periodic_tf32 / mel_weight_matrix_tf32 below: the product's default since
    round 5, mt3_frontend_config.table_dtype) -- independently of `dtype`, the
    arithmetic they are USED in."""
    if tables not in ("float64", "tf32"):
        raise ValueError("tables must be 'float64' or 'tf32'")
    x = np.asarray(samples, dtype)
    hann, melw = (hann_periodic_tf32(), mel_weight_matrix_tf32()) if tables == "tf32" else (hann_periodic(), mel_weight_matrix())
    frames = frame_signal(x) * hann.astype(dtype)[None, :]
    mag = np.abs(np.fft.rfft(frames.astype(np.float64), axis=-1)).astype(dtype)
    mel = mag @ melw.astype(dtype)
    safe = np.where(mel <= 0.0, dtype(LOG_EPS), mel)
    return np.log(safe).astype(dtype)


# ---------------------------------------------------------------------------------------------------------------
# Second variant: the same leaves in FLOAT32 with TensorFlow's own op order (round 4).  tf.signal builds its window and
# its mel matrix in `dtype=tf.float32` by default and the reference passes no dtype (mt3/spectral_ops.py:42-47 `tf.signal
# .stft(... pad_end=True)`, :69-71 `tf.signal.linear_to_mel_weight_matrix(...)`), so what the reference multiplies by is
# the f32-ROUNDED-AT-EVERY-OP version of the tables above.  Restated [from memory of tensorflow/python/ops/signal/
# {window_ops,mel_ops}.py and math_ops.linspace_nd; still PARITY UNPINNED against TF itself] to BOUND what that rounding
# moves in the log domain (tests/test_oracle_frontend_tf32.py; the figure is quoted in DESIGN.md section 4):
#   linspace(start, stop, n)   = concat(start, start + delta * [1 .. n-2], stop), delta = (stop - start) / (n - 1), all f32
#   _hertz_to_mel(f)           = 1127.0 * log(1.0 + f / 700.0)                   (plain log, not log1p; f32)
#   hann_window(N, periodic)   = 0.5 - 0.5 * cos(2 pi * k / N)                   (2 pi as an f32 constant, f32 cos)
# Round 5: `log` and `cos` are taken CORRECTLY ROUNDED to float32 (evaluated in float64, rounded once).  No float32 log at
# hand is: numpy's differs from the correctly rounded value at 10 % of its arguments, glibc's and Eigen's (TensorFlow's) at
# others, and one ulp of a mel value moves a triangle weight by up to 4.5e-5 -- three float32 evaluations of this formula
# land up to 9.1e-5 apart in a weight, MORE than the 6.8e-5 between any of them and the float64 table.  TensorFlow's table
# is one more point of that cloud; the correctly rounded log is its centre and is reproducible anywhere, so the product's
# default tables (mt3_amd/csrc/frontend_tables.h, the same rule in C++) equal these bit for bit.
#   mel = tensordot(|rfft|, W) , log                                              (f32; the FFT itself in f32: scipy.fft keeps
#                                                                                 single precision, numpy.fft would not)
def _f32(x):
    return np.asarray(x, np.float32)


def linspace_tf32(start, stop, n) -> np.ndarray:
    start, stop = np.float32(start), np.float32(stop)
    delta = np.float32((stop - start) / np.float32(n - 1))
    mid = start + delta * np.arange(1, n - 1, dtype=np.float32)
    return np.concatenate([[start], mid.astype(np.float32), [stop]]).astype(np.float32)


def hertz_to_mel_tf32(f):
    arg = (np.float32(1.0) + _f32(f) / np.float32(700.0)).astype(np.float32)
    return (np.float32(1127.0) * np.log(arg.astype(np.float64)).astype(np.float32)).astype(np.float32)


def mel_weight_matrix_tf32(num_mel_bins=NUM_MEL_BINS, num_spectrogram_bins=FFT_SIZE // 2 + 1,
                           sample_rate=SAMPLE_RATE, lo_hz=MEL_LO_HZ, hi_hz=MEL_HI_HZ) -> np.ndarray:
    nyquist = np.float32(sample_rate) / np.float32(2.0)
    lin = linspace_tf32(0.0, nyquist, num_spectrogram_bins)[1:]
    spec_mel = hertz_to_mel_tf32(lin)[:, None]
    edges = linspace_tf32(hertz_to_mel_tf32(lo_hz), hertz_to_mel_tf32(hi_hz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = ((spec_mel - lower) / (center - lower)).astype(np.float32)
    upper_slopes = ((upper - spec_mel) / (upper - center)).astype(np.float32)
    w = np.maximum(np.float32(0.0), np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[1, 0], [0, 0]]).astype(np.float32)


def hann_periodic_tf32(n=FFT_SIZE) -> np.ndarray:
    count = np.arange(n, dtype=np.float32)
    arg = (np.float32(2.0 * np.pi) * count / np.float32(n)).astype(np.float32)      # periodic, even n: divisor n
    return (np.float32(0.5) - np.float32(0.5) * np.cos(arg.astype(np.float64)).astype(np.float32)).astype(np.float32)


def compute_logmel_tf32(samples: np.ndarray) -> np.ndarray:
    """compute_logmel with every leaf as a float32 TensorFlow graph would evaluate it (see above)."""
    import scipy.fft
    x = _f32(samples)
    frames = (frame_signal(x) * hann_periodic_tf32()[None, :]).astype(np.float32)
    spec = scipy.fft.rfft(frames, axis=-1)
    assert spec.dtype == np.complex64
    mag = np.abs(spec).astype(np.float32)
    mel = (mag @ mel_weight_matrix_tf32()).astype(np.float32)
    safe = np.where(mel <= 0.0, np.float32(LOG_EPS), mel)
    return np.log(safe).astype(np.float32)


def segment_logmel_padded(seg_frames: np.ndarray, inputs_length: int, dtype=np.float32) -> np.ndarray:
    """What reaches the encoder for one segment: log-mel of its n<=T frames, rows
    n..T-1 literal zeros (feature converter pads AFTER the log: models.py:48-98)."""
    n = seg_frames.shape[0]
    out = np.zeros((inputs_length, NUM_MEL_BINS), dtype)
    out[:n] = compute_logmel(seg_frames.reshape(-1), dtype=np.float64).astype(dtype)
    return out


def synth_audio(n_segments: int, seed: int = 0, seg_samples: int = 32768) -> np.ndarray:
    """Synthetic 16 kHz audio, SURVEY.md 8(d): per segment 1-6 harmonic tones
    (f0 log-uniform 55..1760 Hz, 8 partials at 1/k) with random on/off inside the
    segment + one 20 ms white-noise burst per 0.25 s; peak-normalised to 0.9."""
    rng = np.random.default_rng(seed)
    t = np.arange(seg_samples) / SAMPLE_RATE
    out = np.zeros((n_segments, seg_samples), np.float32)
    for s in range(n_segments):
        x = np.zeros(seg_samples, np.float64)
        for _ in range(int(rng.integers(1, 7))):
            f0 = float(np.exp(rng.uniform(np.log(55.0), np.log(1760.0))))
            a, b = sorted(rng.uniform(0, t[-1], 2))
            env = ((t >= a) & (t <= b)).astype(np.float64)
            ph = rng.uniform(0, 2 * np.pi)
            for k in range(1, 9):
                if f0 * k < SAMPLE_RATE / 2:
                    x += env * np.sin(2 * np.pi * f0 * k * t + ph * k) / k
        burst = int(0.020 * SAMPLE_RATE)
        for q in range(int(t[-1] / 0.25) + 1):
            st = int((q * 0.25 + rng.uniform(0, 0.2)) * SAMPLE_RATE)
            if st + burst <= seg_samples:
                x[st:st + burst] += rng.standard_normal(burst) * 0.5
        x *= 0.9 / max(np.max(np.abs(x)), 1e-9)
        out[s] = x.astype(np.float32)
    return out
