"""CPU emulation of the engine's encoder in its two reduced-precision formats (test infrastructure): the same data flow
as mt3_engine_encode (mt3_amd/csrc/engine.hip) with dense_dtype = MT3_FP8_E4M3 (fmt "mx8") or plain bf16 operands
(fmt "bf16"), every rounding the device performs performed here at the same place -- bf16 operands of the input projection, MXFP8 (tests/mx8_ref.py) for every encoder GEMM operand (weights with
the pre-norm scale folded in, quantised per output row along K; activations per row along K), bf16 q/k/v, bf16
attention probabilities and output, f32 residual stream -- and everything between the roundings in float64.  What it
does NOT emulate is summation order and the scaled MFMA's product alignment (1-2e-4 of sum |products|), so device and
emulation agree closely, not bit for bit (a rounding decision flips now and then and propagates)."""
import numpy as np
import torch

from oracle.network import sinusoidal_table
from tests import mx8_ref


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.float().to(torch.bfloat16).double()


def _mx8(t: torch.Tensor) -> torch.Tensor:
    """rows [M, K] -> their MXFP8 rounding, float64"""
    q, sc = mx8_ref.quantize(t.float().contiguous())
    return mx8_ref.dequantize(q, sc)


def encode(params, cfg, inputs: np.ndarray, fmt: str = "mx8") -> torch.Tensor:
    """inputs [B, T, input_depth] f32 -> encoder output [B, T, emb] (float64 values of the engine's f32 output).
    fmt: the GEMM operand format of the encoder layers, "mx8" (MXFP8) or "bf16"."""
    _op = {"mx8": _mx8, "bf16": _bf16}[fmt]             # rounding of a GEMM operand (rows along K)

    def _wq(w_in_out: np.ndarray, scale=None) -> torch.Tensor:
        """Flax kernel [in, out] (* pre-norm scale along `in`) -> output-major [out, in], rounded along in, float64"""
        w = torch.from_numpy(np.asarray(w_in_out, np.float32))
        if scale is not None:
            w = w * torch.from_numpy(np.asarray(scale, np.float32))[:, None]   # f32 product, as engine.hip:put_transposed
        return _op(w.T.contiguous())

    p = params
    B, T, _ = inputs.shape
    H, D, emb = cfg.num_heads, cfg.head_dim, cfg.emb_dim
    M = B * T
    pe = torch.from_numpy(sinusoidal_table(2048, emb)).double()
    xin = _bf16(torch.from_numpy(np.asarray(inputs, np.float32)).reshape(M, -1))
    w_in = _bf16(torch.from_numpy(p["encoder/continuous_inputs_projection/kernel"]))
    x = (xin @ w_in + pe[:T].repeat(B, 1)).float().double()                     # the residual stream is f32 in memory
    for i in range(cfg.num_encoder_layers):
        L = f"encoder/layers_{i}"
        s1, s2 = p[L + "/pre_attention_layer_norm/scale"], p[L + "/pre_mlp_layer_norm/scale"]
        # attention block: QKV (fused RMSNorm: 1/rms from the f32 rows, applied after the product) -> bf16
        rs = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6)
        xq = _op(x)
        wqkv = torch.cat([_wq(p[L + f"/attention/{n}/kernel"], s1) for n in ("query", "key", "value")])
        qkv = _bf16((xq @ wqkv.T) * rs).reshape(B, T, 3, H, D)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        s = torch.einsum("bqhd,bkhd->bhqk", q, k)                               # unscaled logits (layers.py:230-234)
        pr = torch.exp(s - s.amax(-1, keepdim=True))
        o = torch.einsum("bhqk,bkhd->bqhd", _bf16(pr), v) / pr.sum(-1).permute(0, 2, 1)[..., None]
        attn = _bf16(o.reshape(M, H * D))
        x = (x + _op(attn) @ _wq(p[L + "/attention/out/kernel"]).T).float().double()
        # MLP block: the GEGLU output exists only in the operand format
        rs = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6)
        xq = _op(x)
        g = (xq @ _wq(p[L + "/mlp/wi_0/kernel"], s2).T) * rs
        lin = (xq @ _wq(p[L + "/mlp/wi_1/kernel"], s2).T) * rs
        h = torch.nn.functional.gelu(g, approximate="tanh") * lin
        x = (x + _op(h) @ _wq(p[L + "/mlp/wo/kernel"]).T).float().double()
    sc = torch.from_numpy(np.asarray(p["encoder/encoder_norm/scale"], np.float32)).double()
    out = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6) * sc
    return out.reshape(B, T, emb)
