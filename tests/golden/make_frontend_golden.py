#!/usr/bin/env python3
"""Generate tests/golden/frontend_golden.npz by running the reference's REAL mt3/spectrograms.py and
mt3/spectral_ops.py (imported unmodified from /root/reference) on tests/golden/tf_standin.py.

Pins the reference's own composition and parameters of the frontend (see tf_standin.py for what is and is not
independent here).  Signals: a short ragged one (pad_end framing), white noise, a 1 kHz tone, silence.
Usage (build container only):  python tests/golden/make_frontend_golden.py
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)

import tf_standin  # noqa: E402

tf_standin.install()
pkg = types.ModuleType("mt3")
pkg.__path__ = [os.path.join(REF, "mt3")]
sys.modules["mt3"] = pkg
from mt3 import spectrograms  # noqa: E402  (the reference file, unmodified; pulls in mt3/spectral_ops.py)


def main():
    rng = np.random.default_rng(11)
    cfg = spectrograms.SpectrogramConfig()
    sigs = {"ragged_1000": rng.uniform(-1, 1, 1000).astype(np.float32),
            "noise_4096": rng.uniform(-1, 1, 4096).astype(np.float32),
            "tone_1khz_3000": np.sin(2 * np.pi * 1000.0 * np.arange(3000) / 16000.0).astype(np.float32),
            "silence_640": np.zeros(640, np.float32)}
    out = {}
    for name, x in sigs.items():
        out["in_" + name] = x
        out["logmel_" + name] = np.asarray(spectrograms.compute_spectrogram(x, cfg), np.float32)
        fr = np.asarray(spectrograms.split_audio(x, cfg))
        out["frames_" + name] = fr.astype(np.float32)
        assert np.array_equal(np.asarray(spectrograms.flatten_frames(fr))[: len(x)], x)
    out["meta"] = np.array([cfg.sample_rate, cfg.hop_width, cfg.num_mel_bins, spectrograms.FFT_SIZE,
                            int(spectrograms.input_depth(cfg))], np.int64)
    out["frames_per_second"] = np.float64(cfg.frames_per_second)
    np.savez_compressed(os.path.join(HERE, "frontend_golden.npz"), **out)
    for k, v in out.items():
        if k.startswith("logmel_"):
            print(k, v.shape, float(v.min()), float(v.max()))
    print("wrote frontend_golden.npz")


if __name__ == "__main__":
    main()
