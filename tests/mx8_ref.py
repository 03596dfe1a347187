"""Emulation of the MXFP8 format of mt3_amd/csrc/gemm_mx8.hip (test infrastructure): OCP e4m3fn elements, one E8M0
scale per 32 consecutive elements of a row, scale = 2^(floor(log2 amax) - 7) so that amax / scale is in [128, 256)
(nothing saturates; e4m3fn tops out at 448), elements rounded to nearest even by torch.float8_e4m3fn."""
import torch


def quantize(x: torch.Tensor):
    """x [rows, K] (any float dtype, K % 32 == 0) -> (uint8 e4m3 bytes [rows, K], uint8 E8M0 [rows, K / 32])."""
    rows, K = x.shape
    xb = x.float().reshape(rows, K // 32, 32)
    amax = xb.abs().amax(-1)
    _, ex = torch.frexp(amax)                                  # amax = m * 2^ex, m in [0.5, 1)  ->  floor(log2) = ex - 1
    byte = torch.where(amax > 0, ex + 119, torch.zeros_like(ex)).clamp(min=0)        # (ex - 1) - 7 + 127
    inv = torch.ldexp(torch.ones_like(amax), (127 - byte).clamp(max=127))           # exact powers of two
    q = (xb * inv[..., None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(rows, K), byte.to(torch.uint8)


def dequantize(q: torch.Tensor, sc: torch.Tensor) -> torch.Tensor:
    """-> float64 [rows, K]"""
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).double().reshape(rows, K // 32, 32)
    return (v * torch.ldexp(torch.ones_like(sc, dtype=torch.float64), sc.int() - 127)[..., None]).reshape(rows, K)


def same_values(qa: torch.Tensor, qb: torch.Tensor) -> bool:
    """e4m3 has +0 / -0: compare as values"""
    return torch.equal(qa.view(torch.float8_e4m3fn).float(), qb.view(torch.float8_e4m3fn).float())
