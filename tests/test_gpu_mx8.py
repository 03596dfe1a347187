"""GPU tests of the MXFP8 dense path (dense_dtype fp8_e4m3: BASELINE configs[4] "fp8 MFMA path"), through the C ABI.

There is no reference counterpart (mt3's DenseGeneral is f32, layers.py:311-360), so the checks are of two kinds:
  exactness  -- the device quantisers equal the torch.float8_e4m3fn emulation of the format bit for bit, and the GEMM
                equals a float64 product of the DEQUANTISED operands up to the arithmetic of the scaled MFMA itself
                (v_mfma_scale_f32_16x16x128_f8f6f4 aligns its 128 products before adding: measured 1.2-1.8e-4 of the
                sum of |products| on random operands, tools/micro/mfma_scale_check.hip; bound used here: 2.5e-4);
  distance   -- the encoder output / step-0 logits of an engine on this path against the f32 CPU oracle, bounds below.
"""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network  # noqa: E402
from tests import mx8_ref  # noqa: E402
from tests.test_gpu_engine import _inputs, _oracle, _params, rel  # noqa: E402

MFMA_TOL = 2.5e-4          # |error| <= MFMA_TOL * sum_k |a_k w_k| (+ the rounding of the output type)


def lib():
    return _lib.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_quantize(x, with_ss=False):
    M, K = x.shape
    q = torch.empty(M, K, device="cuda", dtype=torch.uint8)
    sc = torch.empty(M, K // 32, device="cuda", dtype=torch.uint8)
    ss = torch.empty(M, K // 16, device="cuda", dtype=torch.float32) if with_ss else None
    _lib.check(lib().mt3_op_mx8_quantize(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, M, K, q.data_ptr(),
                                         sc.data_ptr(), ss.data_ptr() if with_ss else None, stream()))
    torch.cuda.synchronize()
    return q, sc, ss


def _rows(M, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, device="cuda", generator=g)
    x *= torch.exp(torch.randn(M, 1, device="cuda", generator=g))              # rows of different magnitude
    x[:, 3::97] *= 30.0                                                         # outlier columns
    return x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_device_quantiser_bit_exact(dtype):
    x = _rows(300, 384, 1).to(dtype).contiguous()
    x[7, 64:96] = 0                                                             # an all-zero block
    q, sc, ss = _dev_quantize(x, with_ss=dtype == torch.float32)
    rq, rsc = mx8_ref.quantize(x)
    assert torch.equal(sc, rsc)
    assert mx8_ref.same_values(q, rq)
    if ss is not None:
        ref = (x.double() ** 2).reshape(300, 24, 16).sum(-1)
        assert float(((ss.double() - ref).abs() / ref.clamp_min(1e-30)).max()) < 1e-6


def _host_quantize(w):
    import ctypes as C
    wn = np.ascontiguousarray(w.cpu().numpy())
    q = np.empty(wn.shape, np.uint8)
    sc = np.empty((wn.shape[0], wn.shape[1] // 32), np.uint8)
    _lib.check(lib().mt3_host_mx8_quantize(wn.ctypes.data_as(C.c_void_p), wn.shape[0], wn.shape[1],
                                           q.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)))
    return torch.from_numpy(q).cuda(), torch.from_numpy(sc).cuda()


def _gemm(aq, asc, wq, wsc, epi, out=None, a_ss=None, seq=0, outs=(None, None, None)):
    M, K = aq.shape
    N = wq.shape[0]
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib().mt3_op_gemm_mx8(aq.data_ptr(), asc.data_ptr(), wq.data_ptr(), wsc.data_ptr(), p(out), M, N, K, epi,
                                     seq, p(a_ss), p(outs[0]), p(outs[1]), p(outs[2]), stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(256, 512, 512), (384, 256, 384), (200, 768, 1024), (130, 256, 2048)])
def test_gemm_mx8_every_epilogue(M, N, K):
    x = _rows(M, K, 2)
    g = torch.Generator(device="cuda").manual_seed(3)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    aq, asc, ss = _dev_quantize(x, with_ss=True)
    wq, wsc = _host_quantize(w)
    assert mx8_ref.same_values(wq, mx8_ref.quantize(w)[0])                      # host and device rule are one rule
    ad, wd = mx8_ref.dequantize(aq, asc), mx8_ref.dequantize(wq, wsc)
    prod, mag = ad @ wd.T, ad.abs() @ wd.abs().T
    norm = K <= 1024
    rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6) if norm else torch.ones(M, 1, device="cuda", dtype=torch.float64)

    # STORE: bf16 [M][N], fused RMSNorm from the partial sums
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(aq, asc, wq, wsc, _lib.EPI_STORE, out=out, a_ss=ss if norm else None)
    want = prod * rs
    assert bool(((out.double() - want).abs() <= want.abs() / 256 + MFMA_TOL * mag * rs).all())

    # HEADS: [2][B][H][seq][64]
    if M % 64 == 0:
        seq, H, B = 64, N // 128, M // 64
        outh = torch.zeros(2, B, H, seq, 64, device="cuda", dtype=torch.bfloat16)
        _gemm(aq, asc, wq, wsc, _lib.EPI_HEADS, out=outh, seq=seq)
        got = outh.permute(1, 3, 0, 2, 4).reshape(M, N).double()               # [B][seq][kv][H][64]
        assert bool(((got - prod).abs() <= prod.abs() / 256 + MFMA_TOL * mag).all())

    # RESID: x += product, and the new rows again as MXFP8 + partial sums
    x0 = torch.randn(M, N, device="cuda", generator=g)
    xr = x0.clone()
    oq = torch.empty(M, N, device="cuda", dtype=torch.uint8)
    osc = torch.empty(M, N // 32, device="cuda", dtype=torch.uint8)
    oss = torch.empty(M, N // 16, device="cuda", dtype=torch.float32)
    _gemm(aq, asc, wq, wsc, _lib.EPI_RESID, out=xr, outs=(oq, osc, oss))
    want = x0.double() + prod
    assert bool(((xr.double() - want).abs() <= want.abs() * 2e-7 + MFMA_TOL * mag + 1e-7).all())
    rq, rsc = mx8_ref.quantize(xr)                                              # of the rows the kernel itself stored
    assert torch.equal(osc, rsc) and mx8_ref.same_values(oq, rq)
    ref = (xr.double() ** 2).reshape(M, N // 16, 16).sum(-1)
    assert float(((oss.double() - ref).abs() / ref.clamp_min(1e-30)).max()) < 1e-6

    # GEGLU: W rows interleaved gate / linear in 16s; the output exists only as MXFP8
    if N % 256 == 0:
        hq = torch.empty(M, N // 2, device="cuda", dtype=torch.uint8)
        hsc = torch.empty(M, N // 64, device="cuda", dtype=torch.uint8)
        _gemm(aq, asc, wq, wsc, _lib.EPI_GEGLU, a_ss=ss if norm else None, outs=(hq, hsc, None))
        p4, m4 = (prod * rs).reshape(M, N // 32, 2, 16), (mag * rs).reshape(M, N // 32, 2, 16)
        gate, lin = p4[:, :, 0], p4[:, :, 1]
        gelu = lambda v: 0.5 * v * (1 + torch.tanh(0.7978845608028654 * (v + 0.044715 * v ** 3)))
        want = (gelu(gate) * lin).reshape(M, N // 2)
        # first-order propagation of the MFMA bound through gelu(g) * l (|gelu'| <= 1.13), then one e4m3 rounding
        eg, el = MFMA_TOL * m4[:, :, 0], MFMA_TOL * m4[:, :, 1]
        slack = (1.13 * eg * lin.abs() + (gelu(gate).abs() + 1.13 * eg) * el).reshape(M, N // 2)
        got = mx8_ref.dequantize(hq, hsc)
        step = torch.ldexp(torch.ones_like(hsc, dtype=torch.float64), hsc.int() - 127).repeat_interleave(32, 1) / 512
        assert bool(((got - want).abs() <= (want.abs() + slack) / 16 + slack + step + 1e-6 * want.abs()).all())
        # and the block scales are those of the exact result wherever that is not within the slack of a binade edge
        assert float((hsc != mx8_ref.quantize(want.float())[1]).float().mean()) < 0.02


def test_gemm_mx8_rejects_what_it_cannot_do():
    z = torch.zeros(128, 128, device="cuda", dtype=torch.uint8)
    s = torch.zeros(128, 4, device="cuda", dtype=torch.uint8)
    o = torch.zeros(128, 128, device="cuda", dtype=torch.bfloat16)
    p = lambda t: t.data_ptr()
    bad = lambda *a: lib().mt3_op_gemm_mx8(*a) != 0
    assert bad(p(z), p(s), p(z), p(s), p(o), 128, 128, 96, _lib.EPI_STORE, 0, None, None, None, None, stream())    # K % 128
    assert bad(p(z), p(s), p(z), p(s), p(o), 128, 64, 128, _lib.EPI_STORE, 0, None, None, None, None, stream())    # N % 128
    assert bad(p(z), p(s), p(z), p(s), p(o), 128, 128, 128, _lib.EPI_RESID, 0, None, None, None, None, stream())   # no MXFP8 outputs
    assert bad(p(z), p(s), p(z), p(s), p(o), 128, 128, 128, _lib.EPI_POS, 0, None, None, None, None, stream())     # epilogue
    assert b"gemm_mx8" in lib().mt3_last_error()


T, L = 256, 1024


@pytest.fixture(scope="module")
def setup():
    cfg32 = network.T5Config(dtype="float32")
    params = _params(cfg32, seed=0, eos_boost=2.5)
    x = _inputs(3, seed=0)
    x[2, 100:] = 0.0
    orc = _oracle(cfg32, params)
    enc_ref = orc.encode(x)
    _, logits_ref = orc.greedy_decode(enc_ref, 2, return_logits=True)
    return dict(params=params, x=x, enc_ref=enc_ref.numpy(), logits_ref=logits_ref.numpy())


@pytest.mark.parametrize("kv", ["", "fp8_e4m3"])
def test_engine_mx8_encoder_within_bounds(setup, kv):
    """Encoder on MXFP8 vs the f32 oracle.  e4m3 keeps 3 mantissa bits (rounding noise ~2.6 % rms per operand), so
    every GEMM output carries ~3.7 % relative noise whatever its K, and 16 of them feed the residual stream: measured
    on the random-init MT3 shape rel-L2 5.6-9.2e-2 per segment, cosine 0.9958-0.9984 (bf16 path: 6-8e-3).
    Bounds: rel-L2 < 1.3e-1, cosine > 0.992; step-0 logits rel-L2 < 1e-1 (bf16 path: 3e-2)."""
    cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16", dense_dtype="fp8_e4m3", kv_dtype=kv)
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=3)
    eng.load_params(setup["params"])
    assert eng.status(_lib.STATUS_DENSE_FP8) == 1
    enc = eng.encode(torch.from_numpy(setup["x"]).cuda(), return_encoded=True).cpu().numpy()
    ref = setup["enc_ref"]
    assert np.isfinite(enc).all()
    for b in range(3):
        r = rel(enc[b], ref[b])
        cos = float((enc[b] * ref[b]).sum() / (np.linalg.norm(enc[b]) * np.linalg.norm(ref[b])))
        print(f"mx8 encoder (kv {kv or 'bf16'}) segment {b}: rel-L2 {r:.3e} cosine {cos:.5f}")
        assert r < 1.3e-1 and cos > 0.992, f"segment {b}: rel-L2 {r}, cosine {cos}"
    ids, logits0 = eng.decode(num_steps=8, return_first_logits=True)
    r = rel(logits0.cpu().numpy(), setup["logits_ref"][:, 0])
    print(f"mx8 encoder (kv {kv or 'bf16'}): step-0 logits rel-L2 {r:.3e}")
    assert r < 1e-1, f"step-0 logits rel-L2 {r}"          # measured 5.1e-2 (round 4: bound tightened from 2e-1)
    # the bf16 engine on the same weights is the nearer neighbour: both engines agree on every comfortable argmax
    ref_eng = network.Transformer(dataclasses.replace(cfg, dense_dtype=""), input_length=T, max_decode_length=L, max_batch=3)
    ref_eng.load_params(setup["params"])
    assert ref_eng.status(_lib.STATUS_DENSE_FP8) == 0
    ref_eng.encode(torch.from_numpy(setup["x"]).cuda())
    _, lb = ref_eng.decode(num_steps=8, return_first_logits=True)
    lb = lb.cpu().numpy()
    top2 = np.sort(lb, -1)[:, -2:]
    l8 = logits0.cpu().numpy()
    print(f"mx8 vs bf16 engine: step-0 logits rel-L2 {rel(l8, lb):.3e}; top-1/top-2 margins / std "
          f"{((top2[:, 1] - top2[:, 0]) / lb.std()).round(3)}; argmax equal {l8.argmax(-1) == lb.argmax(-1)}")
    safe = (top2[:, 1] - top2[:, 0]) > 0.5 * lb.std()
    assert np.array_equal(l8.argmax(-1)[safe], lb.argmax(-1)[safe])


def test_engine_mx8_matches_the_cpu_emulation_of_the_format(setup):
    """The engine's encoder output against tests/mx8_encoder_ref.py, which rounds where the engine rounds (MXFP8
    operands, bf16 q/k/v/probabilities/attention output, f32 residual) and is exact in between.  Not bit-identical and
    not expected to be: summation order and the scaled MFMA's product alignment move values by ~2e-3 of an output, an
    e4m3 rounding step is 6 % of the element, so a few per cent of the roundings flip at every quantisation point and
    each flip propagates.  Measured 3.7e-2 / 3.5e-2 / 5.0e-2 (the short segment) against 6.2e-2 / 5.6e-2 / 9.2e-2 from either to the f32 oracle -- the device sits
    where the format puts it (the CPU test test_mxfp8_encoder_emulation_distance_from_the_f32_oracle reproduces the
    device's own distance to f32 to three digits: 6.195e-2 / 5.631e-2 emulated, 6.161e-2 / 5.644e-2 on the device)."""
    from tests import mx8_encoder_ref
    cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16", dense_dtype="fp8_e4m3")
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=3)
    eng.load_params(setup["params"])
    enc = eng.encode(torch.from_numpy(setup["x"]).cuda(), return_encoded=True).cpu().double()
    emu = mx8_encoder_ref.encode(setup["params"], cfg, setup["x"])
    for b in range(3):
        r = float((enc[b] - emu[b]).norm() / emu[b].norm())
        print(f"mx8 engine vs CPU emulation, segment {b}: rel-L2 {r:.3e}")
        assert r < 7e-2, f"segment {b}: rel-L2 {r}"


def test_engine_bf16_matches_the_cpu_emulation_of_the_format(setup):
    """The same comparison for the bf16 encoder (the benched path): tests/mx8_encoder_ref.py with fmt="bf16" rounds
    operands to bf16 where the engine does.  Measured: engine vs emulation 2.5-3.6e-3; engine vs f32 oracle 4.595e-3 /
    4.298e-3 / 5.997e-3 where the emulation predicts 4.588e-3 / 4.311e-3 / 6.007e-3 -- the engine's distance from the
    reference precision is the format's.  Bound 6e-3 (the engine is held to 2e-2 against f32 elsewhere)."""
    from tests import mx8_encoder_ref
    cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16")
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=3)
    eng.load_params(setup["params"])
    enc = eng.encode(torch.from_numpy(setup["x"]).cuda(), return_encoded=True).cpu().double()
    emu = mx8_encoder_ref.encode(setup["params"], cfg, setup["x"], fmt="bf16")
    ref = torch.from_numpy(setup["enc_ref"]).double()
    for b in range(3):
        r = float((enc[b] - emu[b]).norm() / emu[b].norm())
        print(f"bf16 engine vs CPU emulation, segment {b}: rel-L2 {r:.3e}; engine vs f32 oracle "
              f"{float((enc[b] - ref[b]).norm() / ref[b].norm()):.3e}; emulation vs f32 oracle "
              f"{float((emu[b] - ref[b]).norm() / ref[b].norm()):.3e}")
        assert r < 6e-3, f"segment {b}: rel-L2 {r}"


def test_engine_mx8_is_deterministic_and_batch_independent(setup):
    """Block scales are per row, tiles never mix rows: a segment's result does not depend on its batch neighbours."""
    cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16", dense_dtype="fp8_e4m3")
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=3)
    eng.load_params(setup["params"])
    x = torch.from_numpy(setup["x"]).cuda()
    a = eng.encode(x, return_encoded=True).clone()
    b = eng.encode(x, return_encoded=True).clone()
    assert torch.equal(a, b)
    c = eng.encode(x[[2, 0, 1]].contiguous(), return_encoded=True)
    assert torch.equal(c, a[[2, 0, 1]])


def test_mx8_needs_bf16():
    with pytest.raises(ValueError):
        network.Transformer(dataclasses.replace(network.T5Config(), dtype="float32", dense_dtype="fp8_e4m3"))
