"""The frontend oracle (and the product's host-side framing helpers) against golden vectors produced by the
reference's REAL mt3/spectrograms.py + mt3/spectral_ops.py, run unmodified in the build container on a numpy
stand-in for TensorFlow (tests/golden/make_frontend_golden.py, tf_standin.py).  Pins the reference's composition
and parameters; the tf.signal leaves are restated from the TensorFlow documentation in that stand-in, separately
from oracle/frontend.py (TensorFlow itself cannot be installed: DESIGN.md section 4)."""
import os

import numpy as np
import pytest

from oracle import frontend as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz")
NAMES = ("ragged_1000", "noise_4096", "tone_1khz_3000", "silence_640")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_constants(gold):
    sr, hop, bins, fft, depth = (int(v) for v in gold["meta"])
    assert (sr, hop, bins, fft, depth) == (F.SAMPLE_RATE, F.HOP_WIDTH, F.NUM_MEL_BINS, F.FFT_SIZE, 512)
    assert float(gold["frames_per_second"]) == 125.0


@pytest.mark.parametrize("name", NAMES)
def test_logmel_matches_the_reference_frontend(gold, name):
    x, want = gold["in_" + name], gold["logmel_" + name]
    got = F.compute_logmel(x.astype(np.float64), np.float64)
    assert got.shape == want.shape
    floor = want == np.float32(np.log(1e-5))
    # where the reference hit the 1e-5 floor exactly (empty mel columns, silence) the oracle does too
    empty_cols = np.zeros(512, bool)
    empty_cols[[1, 10]] = True
    assert floor[:, empty_cols].all() and np.all(got[:, empty_cols] == np.log(1e-5))
    if name.startswith("silence"):
        assert floor.all() and np.all(got == np.log(1e-5))
        return
    # linear domain: f32 FFT/window noise of the reference-side run relative to the frame's spectral peak
    a, b = np.exp(got), np.exp(want.astype(np.float64))
    peak = b.max(axis=1, keepdims=True)
    assert np.abs(a - b).max() <= 2e-6 * peak.max(), np.abs(a - b).max() / peak.max()
    strong = b >= 1e-3 * peak
    assert np.abs(got - want)[strong].max() < 2e-4


@pytest.mark.parametrize("name", NAMES)
def test_hop_framing_matches_split_audio(gold, name):
    from mt3_amd import spectrograms as S
    x, want = gold["in_" + name], gold["frames_" + name]
    got = np.asarray(S.split_audio(x, S.SpectrogramConfig()))
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(np.asarray(S.flatten_frames(got))[: len(x)], x)
