"""Audio spectrogram functions (mirror of mt3/spectrograms.py) on the HIP frontend."""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import _lib

DEFAULT_SAMPLE_RATE = 16000
DEFAULT_HOP_WIDTH = 128
DEFAULT_NUM_MEL_BINS = 512
FFT_SIZE = 2048
MEL_LO_HZ = 20.0
MEL_HI_HZ = 7600.0     # spectral_ops.compute_logmel's default hi_hz (spectral_ops.py:79)


@dataclasses.dataclass
class SpectrogramConfig:
    sample_rate: int = DEFAULT_SAMPLE_RATE
    hop_width: int = DEFAULT_HOP_WIDTH
    num_mel_bins: int = DEFAULT_NUM_MEL_BINS

    @property
    def abbrev_str(self):
        """Suffix naming the non-default fields (used by the reference in task names)."""
        tags = (("sr", self.sample_rate, DEFAULT_SAMPLE_RATE), ("hw", self.hop_width, DEFAULT_HOP_WIDTH),
                ("mb", self.num_mel_bins, DEFAULT_NUM_MEL_BINS))
        return "".join("%s%d" % (k, v) for k, v, default in tags if v != default)

    @property
    def frames_per_second(self):
        return self.sample_rate / self.hop_width


_frontends = {}
# Arithmetic the frontend's Hann window and mel matrix are built in (mt3_frontend_config.table_dtype): "float32" = as
# TensorFlow builds them (tf.signal's dtype default, which the reference does not override: spectral_ops.py:42-47,69-71);
# "float64" = rounds 1-4's evaluation.  A module-level setting like the reference's module constants, read when a
# frontend is first created for a configuration.
TABLE_DTYPE = "float32"


def _frontend(cfg: SpectrogramConfig, table_dtype: str = None):
    table_dtype = table_dtype or TABLE_DTYPE
    if table_dtype not in ("float32", "float64"):
        raise ValueError("table_dtype must be 'float32' or 'float64'")
    key = (cfg.sample_rate, cfg.hop_width, cfg.num_mel_bins, table_dtype)
    if key not in _frontends:
        lib = _lib.load()
        fc = _lib.FrontendConfig(cfg.sample_rate, cfg.hop_width, cfg.num_mel_bins, FFT_SIZE, MEL_LO_HZ, MEL_HI_HZ,
                                 0 if table_dtype == "float32" else 1)
        h = C.c_void_p()
        _lib.check(lib.mt3_frontend_create(C.byref(fc), C.byref(h)))
        _frontends[key] = h
    return _frontends[key]


def split_audio(samples, spectrogram_config: SpectrogramConfig):
    """tf.signal.frame(frame_length=hop, frame_step=hop, pad_end=True) -> [ceil(N/hop), hop]."""
    hop = spectrogram_config.hop_width
    x = np.asarray(samples)
    n = -(-len(x) // hop)
    out = np.zeros((n, hop), x.dtype)
    out.reshape(-1)[: len(x)] = x
    return out


def flatten_frames(frames):
    return np.asarray(frames).reshape(-1)


def input_depth(spectrogram_config: SpectrogramConfig):
    return spectrogram_config.num_mel_bins


def mel_matrix(spectrogram_config: SpectrogramConfig = SpectrogramConfig(), table_dtype: str = None) -> np.ndarray:
    """The dense [1025, 512] f32 mel matrix the kernel's band tables were built from."""
    out = np.zeros((FFT_SIZE // 2 + 1, spectrogram_config.num_mel_bins), np.float32)
    nnz = C.c_int64()
    _lib.check(_lib.load().mt3_frontend_mel_matrix(_frontend(spectrogram_config, table_dtype), out.ctypes.data,
                                                   C.byref(nnz)))
    return out


def compute_spectrogram_batch(audio, n_frames, spectrogram_config: SpectrogramConfig = SpectrogramConfig(),
                              table_dtype: str = None):
    """Batched segments.  audio: CUDA f32 [S, F*hop]; n_frames: sequence of S ints (true frame count of
    each segment, <= F) or None.  Returns CUDA f32 [S, F, mel] with rows >= n_frames[s] equal to 0.0
    (the zero padding the reference's feature converter applies after the log)."""
    import torch
    hop = spectrogram_config.hop_width
    a = audio.to(device="cuda", dtype=torch.float32).contiguous()
    S, n = a.shape
    if n % (16 * hop):
        raise ValueError("segment length must be a multiple of 16 hops")
    F = n // hop
    out = torch.empty((S, F, spectrogram_config.num_mel_bins), device="cuda", dtype=torch.float32)
    nf = None
    if n_frames is not None:
        nf = np.ascontiguousarray(np.asarray(n_frames, np.int32))
        if nf.shape != (S,):
            raise ValueError("n_frames must have one entry per segment")
    _lib.check(_lib.load().mt3_frontend_logmel(_frontend(spectrogram_config, table_dtype), a.data_ptr(), S, F,
                                               nf.ctypes.data if nf is not None else None, out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
    return out


def compute_spectrogram(samples, spectrogram_config: SpectrogramConfig = SpectrogramConfig()):
    """One segment's flattened samples (length n*hop) -> numpy f32 [n, mel] (reference signature)."""
    import torch
    hop = spectrogram_config.hop_width
    x = np.asarray(samples, np.float32)
    n = -(-len(x) // hop)
    F = max(16, -(-n // 16) * 16)
    buf = np.zeros((1, F * hop), np.float32)
    buf[0, : len(x)] = x
    out = compute_spectrogram_batch(torch.from_numpy(buf).cuda(), [n], spectrogram_config)
    return out[0, :n].cpu().numpy()
