"""The algebra behind the decoder's folded projections (mt3_amd/csrc/engine.hip: build_q_fold, build_qkv_fold), checked in
float64 on the CPU with random weights -- independent of any kernel:

  T5 LayerNorm is  n(y) = y * rsqrt(mean(y^2) + eps) * scale  (mt3/layers.py:604-621), so a projection of the normed
  row factors as  n(y) @ W = rs(y) * (y @ (scale[:, None] * W))  with the ROW SCALAR rs(y) outside the matrix product;
  the residual update  y_new = y_old + a @ Wr  (mt3/network.py:120,136,150) is linear, hence
      y_new @ W' = y_old @ W' + a @ (Wr @ W')                                  (q-fold, qkv-fold)
  and the decoder's first input row is  Embed(tok) + FixedEmbed[t]  (mt3/network.py:217-226), hence
      y_in(0) @ W' = (E @ W')[tok] + (P @ W')[t]                               (layer-0 table rows).
"""
import numpy as np


def _rs(y, eps=1e-6):
    return 1.0 / np.sqrt((y * y).mean(-1, keepdims=True) + eps)


def test_folded_projections_equal_the_separate_ones():
    rng = np.random.default_rng(0)
    B, emb, hd, mlp, V, P = 5, 64, 48, 96, 40, 32
    s1, s2 = rng.uniform(0.5, 1.5, emb), rng.uniform(0.5, 1.5, emb)            # pre_self / pre_cross norm scales
    Wq, Wk, Wv, Wqx = (rng.standard_normal((emb, hd)) / 8 for _ in range(4))
    Wo, Wo_mlp = rng.standard_normal((hd, emb)) / 7, rng.standard_normal((mlp, emb)) / 10
    # ---- what the reference computes (network.py:104-135): norm, project; residual; norm, project
    y2, h = rng.standard_normal((B, emb)), rng.standard_normal((B, mlp))
    y_in = y2 + h @ Wo_mlp                                                     # previous layer's MLP residual (:150)
    n1 = y_in * _rs(y_in) * s1
    q, k, v = n1 @ Wq, n1 @ Wk, n1 @ Wv
    attn = rng.standard_normal((B, hd))                                        # stands for the self-attention output
    y1 = y_in + attn @ Wo                                                      # (:120)
    qx = (y1 * _rs(y1) * s2) @ Wqx                                             # cross-attention query (:129-135)
    # ---- the folded form: scaled weights, unnormalised products accumulated across launches, 1/rms applied last
    Wext = np.concatenate([s1[:, None] * Wq, s1[:, None] * Wk, s1[:, None] * Wv, s2[:, None] * Wqx], 1)   # [emb, 4 hd]
    two_source = np.concatenate([h, y2], 1) @ np.concatenate([Wo_mlp @ Wext, Wext], 0)   # kEpiResidS tile: K = mlp + emb
    assert np.allclose(two_source, y_in @ Wext, rtol=1e-12, atol=1e-12)
    rs1 = _rs(y_in)
    assert np.allclose(two_source[:, :hd] * rs1, q) and np.allclose(two_source[:, hd:2 * hd] * rs1, k)
    assert np.allclose(two_source[:, 2 * hd:3 * hd] * rs1, v)
    qf = two_source[:, 3 * hd:] + attn @ (Wo @ (s2[:, None] * Wqx))            # + the out-projection launch's share
    assert np.allclose(qf * _rs(y1), qx, rtol=1e-11, atol=1e-12)
    # ---- layer 0: the input row is a sum of two table rows, so its projection is a sum of two projected table rows
    E, Pt = rng.standard_normal((V, emb)), rng.standard_normal((P, emb))
    tok, t = rng.integers(0, V, B), rng.integers(0, P, B)
    y0 = E[tok] + Pt[t]
    assert np.allclose((E @ Wext)[tok] + (Pt @ Wext)[t], y0 @ Wext, rtol=1e-12, atol=1e-12)
    # the per-16-column partial sums the kernels carry add up to the row's sum of squares (emb a multiple of 16)
    parts = (y0.reshape(B, emb // 16, 16) ** 2).sum(-1)
    assert np.allclose(1.0 / np.sqrt(parts.sum(-1, keepdims=True) / emb + 1e-6), _rs(y0))


def test_logits_ride_in_the_last_layers_fold_launch():
    """Last decoder layer (engine.hip: build_qkv_fold, `last`): there is no next layer to project for, so the fold
    launch's extra columns are the logits weights scaled by decoder_norm (mt3/network.py:244-261: decoder_norm, then
    logits_dense; no 1/sqrt(emb) rescale since logits_via_embedding is False, mt3/network.py:41).  The launch writes the
    UNNORMALISED product and the row's per-16-column sums of squares; the token-pick kernel multiplies by 1/rms of the
    final residual row (decode_ops.hip: argmax_step_kernel, LogitScale) before it searches -- the arg-max of the scaled
    row is what the reference takes, and scaling by a positive row scalar cannot change it."""
    rng = np.random.default_rng(1)
    B, emb, mlp, V = 7, 64, 96, 50
    sn = rng.uniform(0.5, 1.5, emb)                                            # decoder_norm scale
    Wl, Wo_mlp = rng.standard_normal((emb, V)) / 8, rng.standard_normal((mlp, emb)) / 10
    y2, h = rng.standard_normal((B, emb)), rng.standard_normal((B, mlp))
    y_out = y2 + h @ Wo_mlp                                                    # last layer's MLP residual (:150)
    logits = (y_out * _rs(y_out) * sn) @ Wl                                    # the reference's two steps
    Wp = sn[:, None] * Wl
    unnorm = np.concatenate([h, y2], 1) @ np.concatenate([Wo_mlp @ Wp, Wp], 0)   # the launch's vocab columns
    parts = (y_out.reshape(B, emb // 16, 16) ** 2).sum(-1)                     # written by the same launch's epilogue
    rs = 1.0 / np.sqrt(parts.sum(-1, keepdims=True) / emb + 1e-6)
    assert np.allclose(unnorm * rs, logits, rtol=1e-11, atol=1e-12)
    assert np.array_equal(np.argmax(unnorm, -1), np.argmax(logits, -1))        # rs > 0: the pick itself needs no scale
