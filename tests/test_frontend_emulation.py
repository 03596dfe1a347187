"""CPU emulation of the HIP frontend kernel's lane program (same frontend_core.h / frontend_tables.h
as the GPU build, compiled for the host with g++) against the numpy oracle.  Verifies the FFT-1024
radix 16x4x16 index algebra, the real-FFT untangle, the band-sparse mel tables and the zero-row rule
without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import frontend as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libfrontend_emul.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "mt3_amd", "csrc"), os.path.join(ROOT, "tests", "host", "frontend_emul.cpp"),
                           "-o", out])
    return ctypes.CDLL(out)


TABLES = ("float64", "tf32")          # mt3_frontend_config.table_dtype 1 / 0 (the default)


def _run(lib, x, n, tables="float64"):
    x = np.ascontiguousarray(x, np.float32)
    out = np.full((256, 512), np.nan, np.float32)
    lib.emul_logmel(x.ctypes.data_as(ctypes.c_void_p), n, 256, out.ctypes.data_as(ctypes.c_void_p), int(tables == "tf32"))
    return out


def test_tables_match_oracle(emul):
    """both constructions of the window and the mel matrix equal the oracle's BIT FOR BIT: the float64 one (rounded once)
    and the float32-in-TensorFlow's-op-order one (every operation one IEEE float32 operation, log / cos correctly rounded:
    no libm can differ)"""
    for tf32, mel_ref, hann_ref in ((0, F.mel_weight_matrix().astype(np.float32), F.hann_periodic().astype(np.float32)),
                                    (1, F.mel_weight_matrix_tf32(), F.hann_periodic_tf32())):
        md = np.zeros((1025, 512), np.float32)
        nnz = emul.emul_mel_dense(md.ctypes.data_as(ctypes.c_void_p), tf32)
        assert nnz == 1934
        np.testing.assert_array_equal(md, mel_ref)
        hw = np.zeros(2048, np.float32)
        assert emul.emul_hann(hw.ctypes.data_as(ctypes.c_void_p), tf32) == 2048
        np.testing.assert_array_equal(hw, hann_ref)
    # how far the two constructions are apart, and how far float32 evaluations of the SAME formula are from each other
    a, b = F.mel_weight_matrix_tf32(), F.mel_weight_matrix().astype(np.float32)
    assert ((a != 0) == (b != 0)).all() and 3e-5 < np.abs(a - b).max() < 1e-4


@pytest.mark.parametrize("tables", TABLES)
@pytest.mark.parametrize("n", [256, 100, 17, 1])
def test_lane_program_matches_oracle(emul, n, tables):
    audio = F.synth_audio(2, seed=n)
    for x in audio:
        got = _run(emul, x, n, tables)
        ref = F.compute_logmel(x[: n * 128], np.float64, tables=tables)
        assert np.all(got[n:] == 0.0)
        hann = F.hann_periodic_tf32().astype(np.float64) if tables == "tf32" else F.hann_periodic()
        frames = F.frame_signal(x[: n * 128].astype(np.float64)) * hann
        peak = np.abs(np.fft.rfft(frames, axis=-1)).max(1)
        lin = np.abs(np.exp(got[:n].astype(np.float64)) - np.exp(ref))
        assert np.all(lin <= 4e-6 * peak[:, None] + 1.1e-10)
        sig = np.exp(ref) >= 1e-3 * np.maximum(peak[:, None], 1e-30)
        if sig.any():
            assert np.abs(got[:n] - ref)[sig].max() < 2e-4
        assert np.all(got[:n, [1, 10]] == np.float32(np.log(np.float32(1e-5))))


def test_white_noise_and_silence(emul):
    x = np.random.default_rng(0).uniform(-1, 1, 32768).astype(np.float32)
    got = _run(emul, x, 256)
    ref = F.compute_logmel(x, np.float64)
    mask = np.ones(512, bool)
    mask[[1, 10]] = False
    assert np.abs(got - ref)[:, mask].max() < 1e-4
    z = _run(emul, np.zeros(32768, np.float32), 256)
    assert np.all(z == np.float32(np.log(np.float32(1e-5))))
