"""GPU parity tests, kernel by kernel, through the C ABI (libmt3hip.so).

Each HIP kernel is compared with a float64 evaluation of the same operation on the
same (already rounded) inputs; tolerances are stated next to each check.  Integer
kernels are bit-exact against the oracle.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib  # noqa: E402

BF16, F32 = _lib.MT3_BF16, _lib.MT3_F32


def lib():
    return _lib.load()


def stream():
    return torch.cuda.current_stream().cuda_stream


def tdt(dtype):
    return torch.bfloat16 if dtype == BF16 else torch.float32


def rel_err(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-30))


def gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


# ----------------------------------------------------------------------------- GEMM
def run_gemm(dtype, A, a_f32, norm, Wt, out, M, N, K, epi, aux=None, seq_len=0, small=False):
    _lib.check(lib().mt3_op_gemm(dtype, A.data_ptr(), int(a_f32), int(norm), Wt.data_ptr(), out.data_ptr(), M, N, K,
                                 epi, aux.data_ptr() if aux is not None else None, seq_len, int(small), stream()))
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("M,N,K", [(256, 384, 512), (200, 1152, 512), (777, 512, 1024), (3, 128, 384)])
def test_gemm_store_and_resid(dtype, small, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    ct = tdt(dtype)
    A = torch.randn(M, K, device="cuda", generator=g).to(ct)
    Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    ref = A.double() @ Wt.double().T
    out = torch.zeros(M, N, device="cuda", dtype=ct)
    run_gemm(dtype, A, False, False, Wt, out, M, N, K, _lib.EPI_STORE, small=small)
    tol = 6e-3 if dtype == BF16 else 2e-5          # bf16: output rounding 2^-9; f32: accumulation order
    assert rel_err(out, ref) < tol, f"STORE rel err {rel_err(out, ref)}"
    # asymmetric inputs make a transposed C-write fail loudly:
    assert (out.double() - ref).abs().max() < 0.05 * ref.abs().max() + 1e-3
    res0 = torch.randn(M, N, device="cuda", generator=g)
    res = res0.clone()
    run_gemm(dtype, A, False, False, Wt, res, M, N, K, _lib.EPI_RESID, small=small)
    assert rel_err(res, res0.double() + ref) < 2e-5, "RESID"
    o32 = torch.zeros(M, N, device="cuda")
    run_gemm(dtype, A, False, False, Wt, o32, M, N, K, _lib.EPI_F32, small=small)
    assert rel_err(o32, ref) < 2e-5, "F32"


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("small", [False, True])
def test_gemm_norm_fused(dtype, small):
    M, N, K = 300, 1152, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    ct = tdt(dtype)
    x = torch.randn(M, K, device="cuda", generator=g) * (1 + 3 * torch.rand(M, 1, device="cuda", generator=g))
    Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    ref = (x.to(ct).double() @ Wt.double().T) * rs            # the kernel rounds x to CT, then scales in f32
    out = torch.zeros(M, N, device="cuda", dtype=ct)
    run_gemm(dtype, x, True, True, Wt, out, M, N, K, _lib.EPI_STORE, small=small)
    assert rel_err(out, ref) < (6e-3 if dtype == BF16 else 2e-5)
    o32 = torch.zeros(M, N, device="cuda")
    run_gemm(dtype, x, True, True, Wt, o32, M, N, K, _lib.EPI_F32, small=small)
    assert rel_err(o32, ref) < 2e-5
    # and against the un-rounded math (what the reference computes in f32): bf16 input rounding only
    true = (x.double() @ Wt.double().T) * rs
    assert rel_err(o32, true) < (8e-3 if dtype == BF16 else 2e-5)


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("small", [False, True])
def test_gemm_geglu(dtype, small):
    M, K, F = 130, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(9)
    ct = tdt(dtype)
    x = torch.randn(M, K, device="cuda", generator=g)
    w0 = (torch.randn(K, F, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    w1 = (torch.randn(K, F, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    # interleave in 16-row groups: rows [32q,32q+16) = gate cols 16q.., rows [32q+16,32q+32) = linear cols 16q..
    Wt = torch.empty(2 * F, K, device="cuda", dtype=ct)
    Wt.view(F // 16, 2, 16, K)[:, 0] = w0.T.reshape(F // 16, 16, K)
    Wt.view(F // 16, 2, 16, K)[:, 1] = w1.T.reshape(F // 16, 16, K)
    rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    xr = x.to(ct).double()
    ref = gelu_tanh((xr @ w0.double()) * rs) * ((xr @ w1.double()) * rs)
    out = torch.zeros(M, F, device="cuda", dtype=ct)
    run_gemm(dtype, x, True, True, Wt, out, M, 2 * F, K, _lib.EPI_GEGLU, small=small)
    assert rel_err(out, ref) < (8e-3 if dtype == BF16 else 3e-5), f"GEGLU rel err {rel_err(out, ref)}"


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("small", [False, True])
def test_gemm_pos_and_heads(dtype, small):
    B, T, K, N = 3, 256, 512, 512
    M = B * T
    g = torch.Generator(device="cuda").manual_seed(11)
    ct = tdt(dtype)
    x = torch.randn(M, K, device="cuda", generator=g)
    Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    pos = torch.randn(T, N, device="cuda", generator=g)
    out = torch.zeros(M, N, device="cuda")
    run_gemm(dtype, x, True, False, Wt, out, M, N, K, _lib.EPI_POS, aux=pos, seq_len=T, small=small)
    ref = x.to(ct).double() @ Wt.double().T + pos.double().repeat(B, 1)
    assert rel_err(out, ref) < 2e-5
    # HEADS: N = 2*H*64 -> [2][B][H][T][64]
    H = 6
    N2 = 2 * H * 64
    A = torch.randn(M, K, device="cuda", generator=g).to(ct)
    W2 = (torch.randn(N2, K, device="cuda", generator=g) / math.sqrt(K)).to(ct)
    o2 = torch.zeros(2, B, H, T, 64, device="cuda", dtype=ct)
    run_gemm(dtype, A, False, False, W2, o2, M, N2, K, _lib.EPI_HEADS, seq_len=T, small=small)
    r2 = (A.double() @ W2.double().T).view(B, T, 2, H, 64).permute(2, 0, 3, 1, 4)
    assert rel_err(o2, r2) < (6e-3 if dtype == BF16 else 2e-5)


# ---------------------------------------------------- split residual stream + LDS-DMA staged tile (bf16, encoder)
def _split(x):
    """f32 rows -> (bf16 copy, per-16-column sums of squares) through mt3_op_residual_split"""
    M, K = x.shape
    ct = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    ss = torch.empty(M, K // 16, device="cuda")
    _lib.check(lib().mt3_op_residual_split(BF16, x.data_ptr(), ct.data_ptr(), ss.data_ptr(), M, K, stream()))
    torch.cuda.synchronize()
    return ct, ss


def run_gemm_ex(A, norm, Wt, out, M, N, K, epi, seq_len=0, small=False, a_ss=None, out_ct=None, out_ss=None):
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib().mt3_op_gemm_ex(BF16, A.data_ptr(), 0, norm, Wt.data_ptr(), out.data_ptr(), M, N, K, epi, None,
                                    seq_len, int(small), p(a_ss), p(out_ct), p(out_ss), stream()))
    torch.cuda.synchronize()


def test_residual_split_exact():
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(300, 512, device="cuda", generator=g) * 3
    ct, ss = _split(x)
    assert torch.equal(ct, x.to(torch.bfloat16))
    ref = (x.double() ** 2).view(300, 32, 16).sum(-1)
    assert float(((ss.double() - ref).abs() / ref).max()) < 1e-6


@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("M,N,K", [(4096, 1152, 512), (777, 512, 512), (256, 2304, 768), (130, 128, 64)])
def test_gemm_norm2_store_from_split_rows(small, M, N, K):
    """RMSNorm-fused GEMM fed by the split rows (bf16 copy + partial sums): big = the LDS-DMA staged tile."""
    if small and K not in (512, 768):
        pytest.skip("decode tiles take K = 512 / 768 with norm 2")
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g) * 4
    Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ct, ss = _split(x)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    run_gemm_ex(ct, 2, Wt, out, M, N, K, _lib.EPI_STORE, small=small, a_ss=ss)
    rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    ref = (ct.double() @ Wt.double().T) * rs
    assert rel_err(out, ref) < 6e-3, rel_err(out, ref)


@pytest.mark.parametrize("M,N,K", [(2048, 512, 384), (1000, 512, 1024), (640, 768, 2048)])
def test_gemm_glds_resid_updates_the_split_stream(M, N, K):
    g = torch.Generator(device="cuda").manual_seed(K)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    x = torch.randn(M, N, device="cuda", generator=g)
    x0 = x.clone()
    ct = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ss = torch.zeros(M, N // 16, device="cuda")
    run_gemm_ex(A, 0, Wt, x, M, N, K, _lib.EPI_RESID, out_ct=ct, out_ss=ss)
    ref = x0.double() + A.double() @ Wt.double().T
    assert rel_err(x, ref) < 2e-5
    assert torch.equal(ct, x.to(torch.bfloat16))                       # the copy is the rounding of the f32 rows
    ssr = (x.double() ** 2).view(M, N // 16, 16).sum(-1)
    assert float(((ss.double() - ssr).abs() / ssr).max()) < 1e-5


def test_gemm_glds_geglu_and_heads():
    M, K, F = 1536, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(M, K, device="cuda", generator=g) * 2
    w0 = (torch.randn(K, F, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    w1 = (torch.randn(K, F, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    Wt = torch.empty(2 * F, K, device="cuda", dtype=torch.bfloat16)
    Wt.view(F // 16, 2, 16, K)[:, 0] = w0.T.reshape(F // 16, 16, K)
    Wt.view(F // 16, 2, 16, K)[:, 1] = w1.T.reshape(F // 16, 16, K)
    ct, ss = _split(x)
    rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    ref = gelu_tanh((ct.double() @ w0.double()) * rs) * ((ct.double() @ w1.double()) * rs)
    out = torch.zeros(M, F, device="cuda", dtype=torch.bfloat16)
    run_gemm_ex(ct, 2, Wt, out, M, 2 * F, K, _lib.EPI_GEGLU, a_ss=ss)
    assert rel_err(out, ref) < 8e-3, rel_err(out, ref)
    # HEADS: N = 2*H*64 -> [2][B][H][T][64]
    B, T, H = 3, 256, 6
    A = torch.randn(B * T, K, device="cuda", generator=g).to(torch.bfloat16)
    W2 = (torch.randn(2 * H * 64, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    o2 = torch.zeros(2, B, H, T, 64, device="cuda", dtype=torch.bfloat16)
    run_gemm_ex(A, 0, W2, o2, B * T, 2 * H * 64, K, _lib.EPI_HEADS, seq_len=T)
    r2 = (A.double() @ W2.double().T).view(B, T, 2, H, 64).permute(2, 0, 3, 1, 4)
    assert rel_err(o2, r2) < 6e-3


def test_gemm_glds_256_row_tile_store_geglu_heads():
    """The 256 x 128 LDS-DMA tile (round 3; taken when a launch has >= 1024 of them) against float64 on the same bf16
    operands, ragged M (the last tile holds 37 rows), every epilogue it serves, K = 512 and K = 1024 (norm-2 partial
    sums in 16 registers).  The same product launched in row slices small enough to fall back to the 128 x 128 tile
    must agree with the reference AND with the tall-tile launch bit for bit (same per-element summation order)."""
    g = torch.Generator(device="cuda").manual_seed(77)
    for (M, N, K, piece) in ((256 * 120 + 37, 1152, 512, 256 * 56), (256 * 64 + 5, 2048, 1024, 256 * 24)):
        assert ((M + 255) // 256) * (N // 128) >= 1024 > ((piece + 255) // 256) * (N // 128)
        x = torch.randn(M, K, device="cuda", generator=g) * torch.rand(M, 1, device="cuda", generator=g) * 4
        Wt = (torch.randn(N, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
        ct, ss = _split(x)
        rs = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6)
        full = (ct.double() @ Wt.double().T) * rs
        F = N // 2
        gate = full.view(M, F // 16, 2, 16)[:, :, 0].reshape(M, F)
        lin = full.view(M, F // 16, 2, 16)[:, :, 1].reshape(M, F)
        for epi, ref, width, tol in ((_lib.EPI_STORE, full, N, 6e-3), (_lib.EPI_GEGLU, gelu_tanh(gate) * lin, F, 8e-3)):
            tall = torch.zeros(M, width, device="cuda", dtype=torch.bfloat16)
            run_gemm_ex(ct, 2, Wt, tall, M, N, K, epi, a_ss=ss)
            assert rel_err(tall, ref) < tol, (M, N, K, epi, rel_err(tall, ref))
            worst = ((tall.double() - ref).abs().amax(1) / ref.abs().amax(1)).max()
            assert float(worst) < 3e-2, float(worst)                     # no row is garbage (ragged tail, tile seams)
            short = torch.zeros(M, width, device="cuda", dtype=torch.bfloat16)
            for r0 in range(0, M, piece):
                m = min(piece, M - r0)
                run_gemm_ex(ct[r0:r0 + m], 2, Wt, short[r0:r0 + m], m, N, K, epi, a_ss=ss[r0:r0 + m])
            assert torch.equal(tall, short), "the 256-row and the 128-row tile must produce identical outputs"
    B, T, H, K = 171, 256, 6, 512                                    # 171 tall tiles x 6 columns = 1026 >= 1024
    A = torch.randn(B * T, K, device="cuda", generator=g).to(torch.bfloat16)
    W2 = (torch.randn(2 * H * 64, K, device="cuda", generator=g) / math.sqrt(K)).to(torch.bfloat16)
    o2 = torch.zeros(2, B, H, T, 64, device="cuda", dtype=torch.bfloat16)
    run_gemm_ex(A, 0, W2, o2, B * T, 2 * H * 64, K, _lib.EPI_HEADS, seq_len=T)
    r2 = (A.double() @ W2.double().T).view(B, T, 2, H, 64).permute(2, 0, 3, 1, 4)
    assert rel_err(o2, r2) < 6e-3
    Bs = 60                                                          # 60 x 6 = 360 tall tiles: the 128-row tile
    o3 = torch.zeros(2, Bs, H, T, 64, device="cuda", dtype=torch.bfloat16)
    run_gemm_ex(A[: Bs * T], 0, W2, o3, Bs * T, 2 * H * 64, K, _lib.EPI_HEADS, seq_len=T)
    assert torch.equal(o3, o2[:, :Bs])


# ------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dtype,T", [(BF16, 256), (BF16, 512), (F32, 256), (F32, 512)])
def test_encoder_attention(dtype, T):
    B, H = 2, 6
    g = torch.Generator(device="cuda").manual_seed(T)
    ct = tdt(dtype)
    qkv = torch.randn(B, T, 3, H, 64, device="cuda", generator=g)
    qkv[:, :, 0] *= 0.35                                   # unscaled logits: keep softmax non-degenerate
    qkv = qkv.to(ct)
    out = torch.zeros(B, T, H * 64, device="cuda", dtype=ct)
    _lib.check(lib().mt3_op_encoder_attention(dtype, qkv.data_ptr(), out.data_ptr(), B, T, H, stream()))
    torch.cuda.synchronize()
    q, k, v = (qkv[:, :, i].double() for i in range(3))
    w = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k), -1)        # no 1/sqrt(d): layers.py:230-234
    ref = torch.einsum("bhqk,bkhd->bqhd", w, v).reshape(B, T, H * 64)
    # bf16: P is rounded to bf16 before P.V (rel 2^-9) and the output is stored in bf16
    assert rel_err(out, ref) < (1e-2 if dtype == BF16 else 3e-5), f"rel err {rel_err(out, ref)}"


@pytest.mark.parametrize("dtype", [BF16, F32])
@pytest.mark.parametrize("n_keys", [1, 2, 31, 257, 1024])
def test_decode_attention_append(dtype, n_keys):
    B, H, cap = 5, 6, 1024
    g = torch.Generator(device="cuda").manual_seed(n_keys)
    ct = tdt(dtype)
    kc = torch.randn(B, H, cap, 64, device="cuda", generator=g).to(ct)
    vc = torch.randn(B, H, cap, 64, device="cuda", generator=g).to(ct)
    qkv = torch.randn(B, 3 * H * 64, device="cuda", generator=g)
    qkv[:, : H * 64] *= 0.35
    qkv = qkv.to(ct)
    step = torch.full((B,), n_keys - 1, device="cuda", dtype=torch.int32)      # per-row position counters
    out = torch.zeros(B, H * 64, device="cuda", dtype=ct)
    kc0, vc0 = kc.clone(), vc.clone()
    es = qkv.element_size()
    _lib.check(lib().mt3_op_decode_attention(dtype, qkv.data_ptr(), 3 * H * 64, kc.data_ptr(), vc.data_ptr(), cap,
                                             qkv.data_ptr() + H * 64 * es, qkv.data_ptr() + 2 * H * 64 * es,
                                             3 * H * 64, step.data_ptr(), 0, out.data_ptr(), B, H, stream()))
    torch.cuda.synchronize()
    q = qkv[:, : H * 64].view(B, H, 64).double()
    kn = qkv[:, H * 64: 2 * H * 64].view(B, H, 64)
    vn = qkv[:, 2 * H * 64:].view(B, H, 64)
    K = kc0.clone()
    V = vc0.clone()
    K[:, :, n_keys - 1] = kn
    V[:, :, n_keys - 1] = vn
    # the cache must now hold the new row, everything else untouched (bit-exact)
    assert torch.equal(kc, K) and torch.equal(vc, V)
    w = torch.softmax(torch.einsum("bhd,bhkd->bhk", q, K[:, :, :n_keys].double()), -1)
    ref = torch.einsum("bhk,bhkd->bhd", w, V[:, :, :n_keys].double()).reshape(B, H * 64)
    assert rel_err(out, ref) < (5e-3 if dtype == BF16 else 2e-5), f"rel err {rel_err(out, ref)}"


@pytest.mark.parametrize("dtype", [BF16, F32])
def test_decode_attention_cross(dtype):
    B, H, T = 4, 6, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    ct = tdt(dtype)
    kv = torch.randn(2, B, H, T, 64, device="cuda", generator=g).to(ct)
    q = (torch.randn(B, H * 64, device="cuda", generator=g) * 0.35).to(ct)
    out = torch.zeros(B, H * 64, device="cuda", dtype=ct)
    _lib.check(lib().mt3_op_decode_attention(dtype, q.data_ptr(), H * 64, kv[0].data_ptr(), kv[1].data_ptr(), T,
                                             None, None, 0, None, T, out.data_ptr(), B, H, stream()))
    torch.cuda.synchronize()
    w = torch.softmax(torch.einsum("bhd,bhkd->bhk", q.view(B, H, 64).double(), kv[0].double()), -1)
    ref = torch.einsum("bhk,bhkd->bhd", w, kv[1].double()).reshape(B, H * 64)
    assert rel_err(out, ref) < (5e-3 if dtype == BF16 else 2e-5)


# ------------------------------------------------------------------ fp8 (e4m3) K/V cache
def _fp8_quant_ref(x):
    """rows [..., 64] (any float dtype) -> (uint8 e4m3fn bytes, power-of-two scale, dequantised f64): the rule of
    fp8_quantize_quad: scale = 2^(exponent(amax) - 7) so that amax / scale is in [128, 256)."""
    x = x.float()
    amax = x.abs().amax(-1, keepdim=True)
    e = torch.frexp(amax)[1].float() - 1                      # amax = m * 2^e, m in [1, 2)
    scale = torch.where(amax > 0, torch.exp2(e - 7), torch.ones_like(amax))
    q = (x / scale).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale.squeeze(-1), q.float().double() * scale.double()


def test_kv_quantize_fp8_bit_exact():
    rows = 2 * 3 * 37
    g = torch.Generator(device="cuda").manual_seed(12)
    src = (torch.randn(2, rows, 64, device="cuda", generator=g) * torch.logspace(-3, 2, rows, device="cuda")[None, :, None])
    src[0, 5] = 0.0                                           # an all-zero row: scale 1, bytes 0
    src = src.to(torch.bfloat16)
    dst = torch.zeros(2, rows, 64, device="cuda", dtype=torch.uint8)
    sc = torch.zeros(rows, 2, device="cuda")
    _lib.check(lib().mt3_op_kv_quantize_fp8(src.data_ptr(), dst.data_ptr(), sc.data_ptr(), rows, stream()))
    torch.cuda.synchronize()
    qk, sk, _ = _fp8_quant_ref(src[0])
    qv, sv, _ = _fp8_quant_ref(src[1])
    assert torch.equal(sc[:, 0], sk) and torch.equal(sc[:, 1], sv)
    # e4m3 has +0 / -0: compare as values
    assert torch.equal(dst[0].view(torch.float8_e4m3fn).float(), qk.view(torch.float8_e4m3fn).float())
    assert torch.equal(dst[1].view(torch.float8_e4m3fn).float(), qv.view(torch.float8_e4m3fn).float())


@pytest.mark.parametrize("n_keys", [1, 2, 17, 49, 257, 1024])
def test_decode_attention_fp8_append(n_keys):
    """fp8 cache: the kernel's result equals exact attention over the DEQUANTISED cache (incl. the row it appends
    and quantises itself); the cache bytes / scales of the new row follow the quantisation rule, others untouched."""
    B, H, cap = 5, 6, 1024
    g = torch.Generator(device="cuda").manual_seed(100 + n_keys)
    kb, ks, kd = _fp8_quant_ref(torch.randn(B, H, cap, 64, device="cuda", generator=g) * 1.3)
    vb, vs, vd = _fp8_quant_ref(torch.randn(B, H, cap, 64, device="cuda", generator=g) * 0.7)
    kc, vc = kb.clone(), vb.clone()
    sc = torch.stack([ks, vs], -1).contiguous()               # [B, H, cap, 2]
    qkv = torch.randn(B, 3 * H * 64, device="cuda", generator=g)
    qkv[:, : H * 64] *= 0.35
    qkv = qkv.to(torch.bfloat16)
    step = torch.full((B,), n_keys - 1, device="cuda", dtype=torch.int32)
    out = torch.zeros(B, H * 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib().mt3_op_decode_attention_fp8(qkv.data_ptr(), 3 * H * 64, kc.data_ptr(), vc.data_ptr(), sc.data_ptr(),
                                                 cap, qkv.data_ptr() + H * 64 * 2, qkv.data_ptr() + 2 * H * 64 * 2,
                                                 3 * H * 64, step.data_ptr(), 0, out.data_ptr(), B, H, stream()))
    torch.cuda.synchronize()
    q = qkv[:, : H * 64].view(B, H, 64).double()
    nkb, nks, nkd = _fp8_quant_ref(qkv[:, H * 64: 2 * H * 64].view(B, H, 64))
    nvb, nvs, nvd = _fp8_quant_ref(qkv[:, 2 * H * 64:].view(B, H, 64))
    pos = n_keys - 1
    f8 = lambda t: t.view(torch.float8_e4m3fn).float()
    assert torch.equal(f8(kc[:, :, pos]), f8(nkb)) and torch.equal(f8(vc[:, :, pos]), f8(nvb))
    assert torch.equal(sc[:, :, pos, 0], nks) and torch.equal(sc[:, :, pos, 1], nvs)
    keep = torch.ones(cap, dtype=torch.bool, device="cuda")
    keep[pos] = False
    assert torch.equal(kc[:, :, keep], kb[:, :, keep]) and torch.equal(vc[:, :, keep], vb[:, :, keep])
    K, V = kd.clone(), vd.clone()
    K[:, :, pos], V[:, :, pos] = nkd, nvd
    w = torch.softmax(torch.einsum("bhd,bhkd->bhk", q, K[:, :, :n_keys]), -1)
    ref = torch.einsum("bhk,bhkd->bhd", w, V[:, :, :n_keys]).reshape(B, H * 64)
    assert rel_err(out, ref) < 5e-3, f"rel err {rel_err(out, ref)}"          # bf16 output rounding only


def test_decode_attention_fp8_cross_and_error_vs_unquantised():
    B, H, T = 4, 6, 256
    g = torch.Generator(device="cuda").manual_seed(4)
    kv = torch.randn(2, B * H * T, 64, device="cuda", generator=g).to(torch.bfloat16)
    q = (torch.randn(B, H * 64, device="cuda", generator=g) * 0.35).to(torch.bfloat16)
    dst = torch.zeros(2, B * H * T, 64, device="cuda", dtype=torch.uint8)
    sc = torch.zeros(B * H * T, 2, device="cuda")
    _lib.check(lib().mt3_op_kv_quantize_fp8(kv.data_ptr(), dst.data_ptr(), sc.data_ptr(), B * H * T, stream()))
    out = torch.zeros(B, H * 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib().mt3_op_decode_attention_fp8(q.data_ptr(), H * 64, dst[0].data_ptr(), dst[1].data_ptr(),
                                                 sc.data_ptr(), T, None, None, 0, None, T, out.data_ptr(), B, H, stream()))
    torch.cuda.synchronize()
    _, _, kd = _fp8_quant_ref(kv[0])
    _, _, vd = _fp8_quant_ref(kv[1])
    qd = q.view(B, H, 64).double()
    att = lambda K, V: torch.einsum("bhk,bhkd->bhd", torch.softmax(torch.einsum(
        "bhd,bhkd->bhk", qd, K.view(B, H, T, 64)), -1), V.view(B, H, T, 64)).reshape(B, H * 64)
    assert rel_err(out, att(kd, vd)) < 5e-3
    # what e4m3 storage costs against the unquantised bf16 cache on UNIT-VARIANCE random rows (3-bit mantissa:
    # 2^-4 relative per element, partly averaged by the dot products; measured 6.0e-2 here, 3.3e-2 on the logits
    # of the whole network, tests/test_gpu_parity_deep.py)
    e = rel_err(out, att(kv[0].double(), kv[1].double()))
    assert e < 9e-2, e


# ------------------------------------------------------------------ ids -> tokens
def test_ids_to_tokens_bit_exact():
    from oracle import symbolic as S
    rng = np.random.default_rng(0)
    L = 1024
    rows = [rng.integers(0, 1536, L), np.zeros(L), np.ones(L), rng.integers(3, 1391, L), rng.integers(0, 3000, L)]
    r = rng.integers(3, 1391, L)
    r[1023] = 1
    rows.append(r)
    r = rng.integers(3, 1391, L)
    r[0] = 1
    rows.append(r)
    ids = np.stack(rows).astype(np.int32)
    vocab = S.GenericTokenVocabulary(1388, extra_ids=100)
    ref = vocab.decode_tf(ids)
    d = torch.from_numpy(ids).cuda()
    out = torch.empty_like(d)
    _lib.check(lib().mt3_ids_to_tokens(d.data_ptr(), ids.shape[0], L, 1388, out.data_ptr(), stream()))
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    # reference literals (vocabularies_test.py:64-83) through the Python mirror
    from mt3_amd import vocabularies as V
    v4 = V.GenericTokenVocabulary(32, extra_ids=4)
    np.testing.assert_array_equal(v4.decode_tf(np.array([0, 2, 3, 4, 34, 35])), [-2, -2, 0, 1, 31, -2])
    v = V.GenericTokenVocabulary(32)
    np.testing.assert_array_equal(v.decode_tf(np.array([0, 2, 3, 4, 1, 0, 1, 0])), [-2, -2, 0, 1, -1, -1, -1, -1])
    assert v.decode([0, 2, 3, 4, 1, 0, 1, 0]) == [-2, -2, 0, 1, -1]
    assert v.decode_tf(np.array([3, 4], np.int64)).dtype == np.int64


# ----------------------------------------------------------------------- frontend
@pytest.mark.parametrize("tables", ["tf32", "float64"])
def test_frontend_vs_oracle(tables):
    """tables: how the Hann window and the mel matrix are BUILT (mt3_frontend_config.table_dtype): "tf32" = float32 in
    TensorFlow's op order, the product's default since round 5 (tf.signal's dtype default, which the reference does not
    override: mt3/spectral_ops.py:42-47,69-71); "float64" = rounds 1-4.  Each against the oracle on the SAME tables, at
    the same bounds: the kernel's distance from the oracle is its f32 FFT, whatever the tables."""
    from oracle import frontend as F
    from mt3_amd import spectrograms as SP
    td = "float32" if tables == "tf32" else "float64"
    audio = F.synth_audio(6, seed=0)
    rng = np.random.default_rng(1)
    audio[4] = rng.uniform(-1, 1, 32768).astype(np.float32)      # white noise: no empty-energy bins
    audio[5] = 0.0                                               # silence: every bin must be log(1e-5)
    n_frames = [256, 256, 100, 1, 256, 256]
    out = SP.compute_spectrogram_batch(torch.from_numpy(audio).cuda(), n_frames, table_dtype=td).cpu().numpy()
    if tables == "tf32":           # the default is the float32 construction
        np.testing.assert_array_equal(out, SP.compute_spectrogram_batch(torch.from_numpy(audio).cuda(), n_frames).cpu().numpy())
    # mel matrix the kernel was built from == the oracle's, bit for bit, in both constructions
    mel_ref_w = F.mel_weight_matrix_tf32() if tables == "tf32" else F.mel_weight_matrix().astype(np.float32)
    np.testing.assert_array_equal(SP.mel_matrix(table_dtype=td), mel_ref_w)
    hann = F.hann_periodic_tf32().astype(np.float64) if tables == "tf32" else F.hann_periodic()
    for s, n in enumerate(n_frames):
        ref = F.compute_logmel(audio[s][: n * 128], np.float64, tables=tables)             # [n, 512]
        assert ref.shape == (n, 512)
        got = out[s]
        assert np.all(got[n:] == 0.0), "pad rows must be exactly 0.0 (not log(eps))"
        frames = F.frame_signal(audio[s][: n * 128].astype(np.float64)) * hann
        peak = np.abs(np.fft.rfft(frames, axis=-1)).max(1)                   # per-frame spectral peak
        mel_ref, mel_got = np.exp(ref), np.exp(got[:n].astype(np.float64))
        # f32 FFT noise floor scales with the frame's peak magnitude: linear-domain bound
        lin = np.abs(mel_got - mel_ref)
        assert np.all(lin <= 4e-6 * peak[:, None] + 1.1e-5 * 1e-5 + 1e-12), f"seg {s}: lin err {lin.max()}"
        # where a bin carries signal (>= 1e-3 of the frame peak) the log itself is tight
        sig = mel_ref >= 1e-3 * np.maximum(peak[:, None], 1e-30)
        if sig.any():
            assert np.abs(got[:n] - ref)[sig].max() < 1e-3
        # the two structurally empty mel columns are exactly log(1e-5)
        assert np.all(got[:n, [1, 10]] == np.float32(np.log(np.float32(1e-5))))
        if tables == "tf32" and n == 256 and s < 2:
            # ... and against the whole-float32 restatement of a TensorFlow graph (f32 FFT as well): both sides now carry
            # f32 FFT noise, so the linear bound doubles
            ref32 = F.compute_logmel_tf32(audio[s]).astype(np.float64)
            assert np.all(np.abs(mel_got - np.exp(ref32)) <= 1.2e-5 * peak[:, None] + 1e-9)
    assert np.all(out[5] == np.float32(np.log(np.float32(1e-5))))
    # linearity property at full size: scaling the audio by 2 adds log 2 to every signal-carrying bin
    big = F.synth_audio(8, seed=3)
    a = SP.compute_spectrogram_batch(torch.from_numpy(big).cuda(), None)
    b = SP.compute_spectrogram_batch(torch.from_numpy(big * 2).cuda(), None)
    mask = a > -4.0
    assert float(((b - a)[mask] - math.log(2.0)).abs().max()) < 2e-3
    # single-segment reference-signature entry point
    one = SP.compute_spectrogram(audio[0][: 100 * 128])
    np.testing.assert_array_equal(one, SP.compute_spectrogram_batch(torch.from_numpy(audio[:1]).cuda(), [100])
                                  .cpu().numpy()[0, :100])


def test_frontend_host_counts_ride_in_the_launch_and_match_device_counts():
    """Round 4: `mt3_frontend_logmel` passes the caller's host frame counts to the kernel by value, 1024 per launch (no
    ring of device slots, no mutex, no device-wide wait): a call with MORE ragged segments than one launch carries is cut
    into pieces, must equal `mt3_frontend_logmel_dev` with the same counts in device memory bit for bit, and works while
    other calls are queued on other streams."""
    from mt3_amd import spectrograms as SP, synthetic
    S, F = 2500, 64                                              # 64-frame segments: 3 launches (1024 + 1024 + 452)
    audio = synthetic.synth_audio(S, seed=5, seg_samples=F * 128)
    rng = np.random.default_rng(2)
    counts = rng.integers(0, F + 1, S).astype(np.int32)
    counts[[0, 1023, 1024, 2047, 2048, S - 1]] = (F, 0, 1, F, 3, F - 1)
    got = SP.compute_spectrogram_batch(audio, counts)
    dev = torch.empty_like(got)
    d_counts = torch.from_numpy(counts).cuda()
    fe = SP._frontend(SP.SpectrogramConfig())
    _lib.check(lib().mt3_frontend_logmel_dev(fe, audio.data_ptr(), S, F, d_counts.data_ptr(), dev.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
    assert torch.equal(got, dev)
    rows = torch.arange(F, device="cuda")[None, :] >= d_counts[:, None]
    assert bool((got[rows] == 0).all()) and bool((got[~rows].abs().sum() > 0))
    # calls from two streams interleave freely: each launch owns its counts
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        a = SP.compute_spectrogram_batch(audio[:1500], counts[:1500])
    with torch.cuda.stream(s2):
        b = SP.compute_spectrogram_batch(audio[1500:], counts[1500:])
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([a, b]), got)
