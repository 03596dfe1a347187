"""Host (weight) side of the MXFP8 dense path, on CPU: mt3_host_mx8_quantize against the torch.float8_e4m3fn emulation
of the format (tests/mx8_ref.py) -- normal values, subnormal e4m3 results, ties, zero blocks, huge and tiny blocks."""
import ctypes as C

import numpy as np
import pytest
import torch

from mt3_amd import _lib
from tests import mx8_ref


def _host(x: np.ndarray):
    rows, K = x.shape
    q = np.empty((rows, K), np.uint8)
    sc = np.empty((rows, K // 32), np.uint8)
    _lib.check(_lib.load().mt3_host_mx8_quantize(x.ctypes.data_as(C.c_void_p), rows, K, q.ctypes.data_as(C.c_void_p),
                                                 sc.ctypes.data_as(C.c_void_p)))
    return torch.from_numpy(q), torch.from_numpy(sc)


def test_host_quantiser_matches_the_e4m3_emulation():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 512, generator=g)
    x[:, ::3] *= torch.exp(3 * torch.randn(64, 171, generator=g))       # wide dynamic range inside the blocks
    x[3, 32:64] = 0.0                                                   # an all-zero block
    x[4, :32] *= 1e30                                                   # a huge block
    x[5, :32] *= 1e-30                                                  # a tiny one
    x[6, 0] = 448.0
    x[6, 1:32] = torch.linspace(0.0, 447.9, 31)
    # exact ties of the 3-bit mantissa (k + 0.5) / 8 at amax 128: must go to even
    x[7, :32] = torch.tensor([128.0] + [(8 + k + 0.5) / 8 for k in range(31)])
    q, sc = _host(np.ascontiguousarray(x.numpy()))
    rq, rsc = mx8_ref.quantize(x)
    assert torch.equal(sc, rsc)
    assert mx8_ref.same_values(q, rq)
    # the format's promise: |x - dequant| <= 2^-4 |x| for e4m3 normals, and nothing saturated
    d = mx8_ref.dequantize(q, sc)
    blk = d.reshape(64, 16, 32).abs().amax(-1) / torch.ldexp(torch.ones(64, 16, dtype=torch.float64), sc.int() - 127)
    assert float(blk.max()) <= 256.0 and float(blk[blk > 0].min()) >= 128.0
    big = (x.abs().double() >= d.reshape(64, 16, 32).abs().amax(-1).repeat_interleave(32, 1) * 2.0 ** -13) & (x != 0)
    assert float(((d - x.double()).abs()[big] / x.abs().double()[big]).max()) <= 2.0 ** -4 + 1e-12


def test_host_quantiser_rejects_ragged_blocks():
    x = np.zeros((2, 48), np.float32)
    with pytest.raises(_lib.Mt3Error):
        _host(x)


def test_host_quantiser_rejects_non_finite_weights():
    x = np.ones((2, 64), np.float32)
    x[1, 40] = np.inf
    with pytest.raises(_lib.Mt3Error, match="non-finite"):
        _host(x)
    x[1, 40] = np.nan
    with pytest.raises(_lib.Mt3Error, match="non-finite"):
        _host(x)


def test_dense_dtype_is_validated_before_anything_touches_the_device():
    import dataclasses
    from mt3_amd import network
    with pytest.raises(ValueError, match="dense_dtype"):
        network.Transformer(dataclasses.replace(network.T5Config(), dtype="float32", dense_dtype="fp8_e4m3"))
    with pytest.raises(ValueError, match="dense_dtype"):
        network.Transformer(dataclasses.replace(network.T5Config(), dtype="bfloat16", dense_dtype="int8"))


def test_mxfp8_encoder_emulation_distance_from_the_f32_oracle():
    """What the FORMAT costs, without a GPU: the CPU emulation of the engine's MXFP8 encoder (tests/mx8_encoder_ref.py)
    against the f32 oracle on the random-init MT3 shape.  The device measures 5.6-9.2e-2 per segment
    (tests/test_gpu_mx8.py); the emulation must land in the same place, or the device is doing something else."""
    from mt3_amd import network
    from oracle import frontend as OF
    from oracle import network as ON
    from tests import mx8_encoder_ref
    cfg = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg, seed=0, norm_scale_jitter=0.2)
    audio = OF.synth_audio(2, seed=0)
    x = np.stack([OF.compute_logmel(a, np.float64).astype(np.float32) for a in audio])
    ref = ON.Oracle(params, ON.T5Config()).encode(x).double()
    emu = mx8_encoder_ref.encode(params, cfg, x)
    emu16 = mx8_encoder_ref.encode(params, cfg, x, fmt="bf16")
    for b in range(2):
        r = float((emu[b] - ref[b]).norm() / ref[b].norm())
        cos = float((emu[b] * ref[b]).sum() / (emu[b].norm() * ref[b].norm()))
        r16 = float((emu16[b] - ref[b]).norm() / ref[b].norm())
        print(f"segment {b}: MXFP8 emulation vs f32 oracle rel-L2 {r:.3e} cosine {cos:.5f}; bf16 emulation {r16:.3e}")
        assert 2e-2 < r < 1.3e-1 and cos > 0.992
        assert 2e-3 < r16 < 1e-2                    # the bf16 engine measures 4.595e-3 / 4.298e-3 on these two segments
