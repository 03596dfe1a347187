"""Second opinions on the unpinned parts of the oracle: TensorFlow / JAX / t5x cannot be installed here, so the
frontend and the network pieces that no reference literal covers are compared with INDEPENDENT implementations
of the same published operators that do exist in this image (torch.stft, scipy's window, torch's tanh-GELU,
rms_norm, scaled_dot_product_attention, a hand-rolled HTK filterbank written from the tf.signal documentation).
This does not replace pinning against the reference, it only rules out restatement slips."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import frontend as F  # noqa: E402
from oracle import network as N   # noqa: E402


def test_stft_magnitude_matches_torch_stft():
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, 5000)
    frames = F.frame_signal(x)                                     # [ceil(5000/128), 2048], zero pad_end
    mag = np.abs(np.fft.rfft(frames * F.hann_periodic(), axis=-1))
    n = frames.shape[0]
    padded = np.concatenate([x, np.zeros((n - 1) * 128 + 2048 - len(x))])
    ref = torch.stft(torch.from_numpy(padded), n_fft=2048, hop_length=128, win_length=2048,
                     window=torch.hann_window(2048, periodic=True, dtype=torch.float64), center=False,
                     return_complex=True).abs().T.numpy()
    assert ref.shape == mag.shape == (n, 1025)
    np.testing.assert_allclose(mag, ref, atol=1e-9)


def test_window_matches_scipy_periodic_hann():
    from scipy.signal import get_window
    np.testing.assert_allclose(F.hann_periodic(), get_window("hann", 2048, fftbins=True), atol=1e-15)


def test_mel_matrix_matches_a_from_the_docs_restatement():
    """tf.signal.linear_to_mel_weight_matrix, as documented: HTK mel = 1127 ln(1 + f/700); linear-frequency bins
    (DC dropped) are mapped to mel; triangle j rises from edge j to edge j+1 and falls to edge j+2, edges
    linspace(mel(lo), mel(hi), bins + 2); weight = max(0, min(rise, fall))."""
    nb, ns, sr, lo, hi = 512, 1025, 16000.0, 20.0, 7600.0
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    edges = [mel(lo) + (mel(hi) - mel(lo)) * i / (nb + 1) for i in range(nb + 2)]
    want = np.zeros((ns, nb))
    for k in range(1, ns):
        m = mel(k * (sr / 2.0) / (ns - 1))
        for j in range(nb):
            if edges[j] < m < edges[j + 2]:
                rise = (m - edges[j]) / (edges[j + 1] - edges[j])
                fall = (edges[j + 2] - m) / (edges[j + 2] - edges[j + 1])
                want[k, j] = max(0.0, min(rise, fall))
    got = F.mel_weight_matrix()
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_gated_gelu_rmsnorm_attention_match_torch_operators():
    torch.manual_seed(0)
    x = torch.randn(5, 7, 64, dtype=torch.float64)
    wi0, wi1, wo = (torch.randn(64, 96, dtype=torch.float64), torch.randn(64, 96, dtype=torch.float64),
                    torch.randn(96, 64, dtype=torch.float64))
    want = (torch.nn.functional.gelu(x @ wi0, approximate="tanh") * (x @ wi1)) @ wo        # T5.1.1 gated-GELU
    np.testing.assert_allclose(N.mlp_block(x, [wi0, wi1], wo).numpy(), want.numpy(), rtol=1e-12, atol=1e-12)
    scale = torch.rand(64, dtype=torch.float64) + 0.5
    want = torch.nn.functional.rms_norm(x, (64,), weight=scale, eps=1e-6)
    np.testing.assert_allclose(N.rms_norm(x, scale).numpy(), want.numpy(), rtol=1e-12, atol=1e-12)
    # attention: [B, T, H, D] in the oracle (Flax layout), [B, H, T, D] in torch; NO 1/sqrt(d) scaling
    q, k, v = (torch.randn(2, 9, 3, 16, dtype=torch.float64) for _ in range(3))
    want = torch.nn.functional.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                                            scale=1.0).transpose(1, 2)
    np.testing.assert_allclose(N.attention(q, k, v).numpy(), want.numpy(), rtol=1e-10, atol=1e-12)


def test_sinusoid_table_matches_the_closed_form():
    """layers.sinusoidal(min_scale=1, max_scale=10000): pe[p, i] = sin(p * s_i), pe[p, i + F/2] = cos(p * s_i),
    s_i = 10000^(-i / (F/2 - 1))."""
    t = N.sinusoidal_table(16, 8)
    for p in (0, 1, 7, 15):
        for i in range(4):
            s = 10000.0 ** (-i / 3.0)
            assert abs(t[p, i] - math.sin(p * s)) < 1e-6 and abs(t[p, i + 4] - math.cos(p * s)) < 1e-6
