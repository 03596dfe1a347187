"""Oracle frontend (numpy): structure quoted in SURVEY.md A.2 and size-independent properties.
PARITY UNPINNED against tf.signal itself (not installable); see oracle/frontend.py."""
import numpy as np

from oracle import frontend as F


def test_mel_matrix_structure():
    w = F.mel_weight_matrix().astype(np.float32)
    nz = w != 0
    assert w.shape == (1025, 512) and nz.sum() == 1934
    assert nz.sum(1).max() == 2 and nz.sum(0).max() == 10
    assert list(np.where(nz.sum(0) == 0)[0]) == [1, 10]
    assert not nz[0].any()                                        # DC row zero


def test_framing_and_window():
    x = np.arange(300, dtype=np.float64)
    fr = F.frame_signal(x)
    assert fr.shape == (3, 2048)                                  # ceil(300/128), pad_end
    assert fr[2, 0] == 256 and fr[2, 43] == 299 and fr[2, 44] == 0 and fr[0, 299] == 299 and fr[0, 300] == 0
    h = F.hann_periodic()
    assert h[0] == 0 and abs(h[1024] - 1) < 1e-15 and abs(h[1] - h[2047]) < 1e-15


def test_logmel_properties():
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, 32768)
    a = F.compute_logmel(x, np.float64)
    assert a.shape == (256, 512)
    np.testing.assert_array_equal(a[:, [1, 10]], np.log(1e-5))     # empty mel columns: exactly log(eps)
    b = F.compute_logmel(2 * x, np.float64)
    mask = np.ones(512, bool)
    mask[[1, 10]] = False
    np.testing.assert_allclose((b - a)[:, mask], np.log(2.0), atol=1e-9)   # linear before the log
    z = F.compute_logmel(np.zeros(1280), np.float64)
    assert z.shape == (10, 512) and np.all(z == np.log(1e-5))
    p = F.segment_logmel_padded(x[: 100 * 128].reshape(100, 128).astype(np.float32), 256)
    assert p.shape == (256, 512) and np.all(p[100:] == 0.0)       # pad rows literal zeros (F8)
    # f32 vs f64 noise floor on signal-carrying bins
    a32 = F.compute_logmel(x.astype(np.float32), np.float32)
    assert np.abs(a32 - a)[:, mask].max() < 1e-4


def test_pure_tone_lands_in_expected_mel_bin():
    t = np.arange(32768) / 16000.0
    x = np.sin(2 * np.pi * 1000.0 * t)
    lm = F.compute_logmel(x, np.float64)[100]
    edges = np.linspace(F.hertz_to_mel(20.0), F.hertz_to_mel(7600.0), 514)
    j = int(np.searchsorted(edges, F.hertz_to_mel(1000.0))) - 1      # triangle whose centre edge follows
    assert abs(int(np.argmax(lm)) - j) <= 1
