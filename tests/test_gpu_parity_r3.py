"""GPU parity cases added in round 3 ("next round" #1b and #7), all through the C ABI:

  (i)   BASELINE configs[4] AS BENCHED: ismir2022/base.gin shape (mt3/gin/ismir2022/base.gin:4-10) with the MXFP8
        encoder dense layers AND the e4m3 K/V caches together, B = 4, teacher-forced logits at 160 cache positions
        against the f32 oracle (reference semantics: mt3/network.py:303-361, mt3/layers.py:246-314);
  (ii)  the MXFP8 engine (MT3 shape) teacher-forced at ALL 1024 cache positions: the MXFP8 cross-K/V at depth;
  (iii) the ismir2021 preset (T = 512) on the f32 engine at the reference's precision: encoder 1e-4, 40 greedy
        steps token-exact (the f32 split-key encoder attention kernel inside the engine);
  (iv)  stale cache contents cannot leak: NaN-poisoned K/V caches (every cache format) give identical ids;
  (v)   a bf16 engine wider than the decode tile's partial-sum registers (emb 1024) encodes at batch 1.
"""
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network  # noqa: E402
from oracle import frontend as OF  # noqa: E402
from oracle import network as ON  # noqa: E402

L, V = 1024, 1536


def _inputs(B, seed, T=256):
    audio = OF.synth_audio(B * (T // 256), seed=seed).reshape(B, -1)
    return np.stack([OF.compute_logmel(a, np.float32) for a in audio])


def _oracle(cfg, params):
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    oc = ON.T5Config(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_heads=cfg.num_heads,
                     num_encoder_layers=cfg.num_encoder_layers, num_decoder_layers=cfg.num_decoder_layers,
                     mlp_dim=cfg.mlp_dim)
    return ON.Oracle(params, oc)


def _rel_rows(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-30)


def _forced(B, S, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(3, 3 + 1388, size=(B, S)).astype(np.int32)


def _teacher_forced_ref(orc, x, forced):
    with torch.no_grad():
        enc = orc.encode(x)
        dec_in = np.concatenate([np.zeros((len(x), 1), np.int32), forced[:, :-1]], 1)
        return enc.numpy(), np.ascontiguousarray(orc.decode_logits(enc, dec_in).numpy().transpose(1, 0, 2))


def test_configs4_as_benched_base_shape_mxfp8_and_fp8_caches_together():
    """What `bench.py`'s `extra.configs4` line times: base.gin shape + dense_dtype fp8 + kv_dtype fp8.  Bounds: round 3
    stated them before measuring (encoder 1.5e-1, logits 2.5e-1) and measured encoder 6.5-9.6e-2, logits max 7.6e-2 / mean
    6.4e-2; round 4 holds the test to what is measured plus a margin for other seeds -- encoder rel-L2
    < 1.2e-1 per segment, teacher-forced logits rel-L2 < 1e-1 at every one of 160 positions, mean < 8e-2 -- so that a
    regression of the MXFP8 / e4m3 path fails it; no drift with depth; the bf16 engine of the same shape is the nearer
    neighbour (printed)."""
    base = dataclasses.replace(network.MT3_BASE, dtype="bfloat16")
    params = network.init_random_params(base, seed=2, norm_scale_jitter=0.1)
    B, S = 4, 160
    x = _inputs(B, seed=6)
    x[3, 120:] = 0.0
    forced = _forced(B, S, 11)
    enc_ref, ref = _teacher_forced_ref(_oracle(base, params), x, forced)
    out = {}
    for name, kv, dense in (("bf16", "", ""), ("configs4", "fp8_e4m3", "fp8_e4m3")):
        cfg = dataclasses.replace(base, kv_dtype=kv, dense_dtype=dense)
        eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B)
        eng.load_params(params)
        enc = eng.encode(torch.from_numpy(x).cuda(), return_encoded=True).cpu().numpy()
        _, logits = eng.decode_forced(forced, num_steps=S)
        out[name] = (enc, logits.cpu().numpy())
        if dense:
            assert eng.status(_lib.STATUS_DENSE_FP8) == 1 and eng.status(_lib.STATUS_KV_FP8) == 1
        assert eng.status(_lib.STATUS_GRAPH_FALLBACKS) == 0
        del eng
    enc, logits = out["configs4"]
    assert np.isfinite(enc).all() and np.isfinite(logits).all()
    re = [float(np.linalg.norm(enc[b] - enc_ref[b]) / np.linalg.norm(enc_ref[b])) for b in range(B)]
    r = _rel_rows(logits, ref)
    r16 = _rel_rows(out["bf16"][1], ref)
    print(f"configs[4] as benched (base.gin + MXFP8 dense + e4m3 caches), B={B}, {S} positions: encoder rel-L2 "
          f"{np.round(re, 4)}; teacher-forced logits vs f32 oracle max {r.max():.3e} mean {r.mean():.3e} "
          f"(bf16 engine of the same shape: max {r16.max():.3e} mean {r16.mean():.3e})")
    assert max(re) < 1.2e-1, re
    assert r.max() < 1e-1 and r.mean() < 8e-2, (r.max(), r.mean())
    assert r[-32:].mean() < 1.5 * r[:32].mean() + 1e-2, (r[:32].mean(), r[-32:].mean())
    top2 = np.partition(ref, -2, axis=-1)[..., -2:]
    safe = (top2[..., 1] - top2[..., 0]) > 0.6 * ref.std(-1)
    assert safe.any()
    assert np.array_equal(logits.argmax(-1)[safe], ref.argmax(-1)[safe])


def test_mxfp8_engine_teacher_forced_all_1024_positions():
    """MT3 shape, dense_dtype fp8 (MXFP8 encoder + MXFP8 cross-K/V projections), bf16 and e4m3 caches: logits at ALL
    1024 cache positions vs the f32 oracle.  The decoder's own dense layers are bf16; what this checks at depth is
    that the MXFP8-made `encoded` and cross-K/V do not make the error grow with the cache position.  Bounds: 1.4e-1 at
    every (step, row), mean 9e-2 (round 4: what is measured plus a margin -- the stated-in-advance 2e-1 could not fail), last 64 positions no worse than 1.5x the first 64."""
    cfg32 = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg32, seed=0, norm_scale_jitter=0.2)
    B = 4
    x = _inputs(B, seed=21)
    x[2, 77:] = 0.0
    forced = _forced(B, L, 5)
    forced[1, 300:] = 0
    _, ref = _teacher_forced_ref(_oracle(cfg32, params), x, forced)
    for kv in ("", "fp8_e4m3"):
        cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16", dense_dtype="fp8_e4m3", kv_dtype=kv)
        eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B)
        eng.load_params(params)
        eng.encode(torch.from_numpy(x).cuda())
        _, logits = eng.decode_forced(forced)
        logits = logits.cpu().numpy()
        r = _rel_rows(logits, ref)
        agree = float((logits.argmax(-1) == ref.argmax(-1)).mean())
        print(f"MXFP8 engine (kv {kv or 'bf16'}), teacher-forced, {B} x 1024 positions: rel-L2 max {r.max():.3e} mean "
              f"{r.mean():.3e}; first/last 64: {r[:64].mean():.3e} / {r[-64:].mean():.3e}; arg-max agreement {agree:.4f}")
        assert r.max() < 1.4e-1 and r.mean() < 9e-2, (r.max(), r.mean())     # measured (r3): max 1.04e-1, mean 6.7e-2
        assert r[-64:].mean() < 1.5 * r[:64].mean() + 5e-3
        del eng


def test_ismir2021_t512_f32_at_reference_precision():
    """ismir2021 preset (T = 512 encoder frames, vocab 1664: NB:176-179) on the f32 engine -- the default of the
    drop-in InferenceModel -- against the f32 oracle at the SAME-precision bounds: encoder rel-L2 < 1e-4 per segment
    (enc_attn_split_kernel<float,512,2,4> inside the engine), step-0 logits < 1e-4, 40 greedy steps token-exact."""
    cfg = dataclasses.replace(network.T5Config(dtype="float32"), vocab_size=1664)
    params = network.init_random_params(cfg, seed=1, norm_scale_jitter=0.1)
    k = params["decoder/logits_dense/kernel"].copy()
    k[:, 1] *= 2.0
    params["decoder/logits_dense/kernel"] = k
    B, S = 3, 40
    x = _inputs(B, seed=9, T=512)
    x[2, 300:] = 0.0
    orc = _oracle(cfg, params)
    with torch.no_grad():
        enc_ref = orc.encode(x)
        ids_ref, logits_ref = orc.greedy_decode(enc_ref, S, return_logits=True)
    enc_ref, logits_ref = enc_ref.numpy(), logits_ref.numpy()
    eng = network.Transformer(cfg, input_length=512, max_decode_length=L, max_batch=B)
    eng.load_params(params)
    enc = eng.encode(torch.from_numpy(x).cuda(), return_encoded=True).cpu().numpy()
    for b in range(B):
        r = np.linalg.norm(enc[b] - enc_ref[b]) / np.linalg.norm(enc_ref[b])
        assert r < 1e-4, f"segment {b}: encoder rel-L2 {r}"
    ids, logits0 = eng.decode(num_steps=S, return_first_logits=True)
    r0 = _rel_rows(logits0.cpu().numpy(), logits_ref[:, 0])
    assert r0.max() < 1e-4, r0
    ids = ids.cpu().numpy()[:, :S]
    for b in range(B):
        if not np.array_equal(ids[b], ids_ref[b]):
            t = int(np.argmax(ids[b] != ids_ref[b]))
            row = logits_ref[b, t]
            top2 = np.partition(row, -2)[-2:]
            gap = float(top2[1] - top2[0]) / float(row.std())
            assert gap < 2e-4, f"row {b} diverges at step {t} although the oracle's margin is {gap:.2e} sigma"
    print(f"ismir2021 (T=512) f32: encoder < 1e-4, step-0 logits {r0.max():.2e}, "
          f"{int((ids == ids_ref).all(1).sum())}/{B} rows token-exact over {S} steps")


@pytest.mark.parametrize("dtype,kv", [("bfloat16", ""), ("float32", ""), ("bfloat16", "fp8_e4m3")])
def test_poisoned_caches_do_not_leak_into_results(dtype, kv):
    """The decode-attention kernels request their first key group before they know the row's length.  What they
    fetched from beyond it must be discarded by position, not multiplied by a zero weight: with every byte of the
    K/V caches (and of the fp8 scale arrays) set to 0xFF -- NaN in bf16, f32 and e4m3 -- before the encode, greedy ids,
    teacher-forced logits and a second, shorter decode over the rows a longer one left behind are unchanged."""
    cfg = dataclasses.replace(network.T5Config(num_encoder_layers=2, num_decoder_layers=3), dtype=dtype, kv_dtype=kv)
    params = network.init_random_params(cfg, seed=4, norm_scale_jitter=0.1)
    B = 5
    x = torch.from_numpy(_inputs(B, seed=3)).cuda()
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B + 2)
    eng.load_params(params)
    eng.encode(x)
    clean_ids, clean_l0 = eng.decode(num_steps=130, return_first_logits=True)
    forced = _forced(B, 70, 8)
    _, clean_tf = eng.decode_forced(forced, num_steps=70)
    clean_ids, clean_l0, clean_tf = clean_ids.clone(), clean_l0.clone(), clean_tf.clone()
    for pattern in (0xFF, 0x7F):                           # NaN patterns (0x7F7F.. is a huge finite bf16 / NaN e4m3)
        eng.debug_poison_caches(pattern, cross=True)
        eng.encode(x)
        ids, l0 = eng.decode(num_steps=130, return_first_logits=True)
        assert torch.isfinite(l0).all()
        assert torch.equal(ids, clean_ids) and torch.equal(l0, clean_l0)
        eng.debug_poison_caches(pattern, cross=False)      # self caches only: the cross rows of this batch stay
        _, tf = eng.decode_forced(forced, num_steps=70)
        assert torch.equal(tf, clean_tf)


def test_kernel_decode_attention_ignores_what_lies_past_the_row():
    """mt3_op_decode_attention / _fp8 on caller-owned caches whose rows past n_keys hold NaN / Inf bit patterns
    (what torch.empty may hand out): same output as over zero-filled tails, bit for bit."""
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    B, H, cap = 7, 6, 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    for kind in ("bf16", "f32", "fp8"):
        tdt = torch.float32 if kind == "f32" else torch.bfloat16
        es = 4 if kind == "f32" else 2
        qkv = (torch.randn(B, 3 * H * 64, device="cuda", generator=g) * 0.3).to(tdt)
        step = torch.tensor([0, 1, 5, 95, 96, 200, 700], device="cuda", dtype=torch.int32)
        outs = []
        for tail in (0.0, float("nan"), float("inf")):
            if kind == "fp8":
                kc = torch.randint(0, 120, (B, H, cap, 64), device="cuda", dtype=torch.uint8,
                                   generator=torch.Generator(device="cuda").manual_seed(1))
                vc = torch.randint(0, 120, (B, H, cap, 64), device="cuda", dtype=torch.uint8,
                                   generator=torch.Generator(device="cuda").manual_seed(2))
                sc = torch.full((B, H, cap, 2), 2.0 ** -7, device="cuda")
                for b in range(B):
                    n = int(step[b])
                    if tail != 0.0:
                        kc[b, :, n:] = 0x7F if np.isnan(tail) else 0xFF          # e4m3fn NaN patterns
                        vc[b, :, n:] = 0xFF if np.isnan(tail) else 0x7F
                        sc[b, :, n:] = tail
                    else:
                        kc[b, :, n:] = 0
                        vc[b, :, n:] = 0
                        sc[b, :, n:] = 0.0
                out = torch.empty(B, H * 64, device="cuda", dtype=torch.bfloat16)
                _lib.check(lib.mt3_op_decode_attention_fp8(qkv.data_ptr(), 3 * H * 64, kc.data_ptr(), vc.data_ptr(),
                                                           sc.data_ptr(), cap, qkv.data_ptr() + H * 64 * es,
                                                           qkv.data_ptr() + 2 * H * 64 * es, 3 * H * 64,
                                                           step.data_ptr(), 0, out.data_ptr(), B, H, s))
            else:
                gk = torch.Generator(device="cuda").manual_seed(1)
                kc = torch.randn(B, H, cap, 64, device="cuda", generator=gk).to(tdt)
                vc = torch.randn(B, H, cap, 64, device="cuda", generator=gk).to(tdt)
                for b in range(B):
                    kc[b, :, int(step[b]):] = tail
                    vc[b, :, int(step[b]):] = -tail if tail == tail else tail
                out = torch.empty(B, H * 64, device="cuda", dtype=tdt)
                _lib.check(lib.mt3_op_decode_attention(_lib.MT3_F32 if kind == "f32" else _lib.MT3_BF16, qkv.data_ptr(),
                                                       3 * H * 64, kc.data_ptr(), vc.data_ptr(), cap,
                                                       qkv.data_ptr() + H * 64 * es, qkv.data_ptr() + 2 * H * 64 * es,
                                                       3 * H * 64, step.data_ptr(), 0, out.data_ptr(), B, H, s))
            torch.cuda.synchronize()
            outs.append(out.clone())
        assert torch.isfinite(outs[0].float()).all(), kind
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), kind


def test_wide_bf16_engine_encodes_at_batch_one():
    """emb 1024 (> the 512 / 768 the decode-sized tile keeps partial sums for): a batch below 2048 rows selects that
    tile, so the encoder must fall back to the single f32 residual stream there instead of failing with
    MT3_ERR_INVALID (engine.hip:719); a batch of 8 (2048 rows) takes the LDS-DMA tile with the split form.
    Both against the oracle at the bf16 bound."""
    cfg = dataclasses.replace(network.T5Config(dtype="bfloat16"), emb_dim=1024, num_heads=8, mlp_dim=2048,
                              num_encoder_layers=2, num_decoder_layers=2)
    params = network.init_random_params(cfg, seed=7, norm_scale_jitter=0.1)
    x = _inputs(8, seed=12)
    orc = _oracle(cfg, params)
    with torch.no_grad():
        enc_ref = orc.encode(x).numpy()
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=8)
    eng.load_params(params)
    for nb in (1, 8):
        enc = eng.encode(torch.from_numpy(x[:nb]).cuda(), return_encoded=True).cpu().numpy()
        for b in range(nb):
            r = np.linalg.norm(enc[b] - enc_ref[b]) / np.linalg.norm(enc_ref[b])
            assert r < 2e-2, (nb, b, r)
        ids = eng.decode(num_steps=6).cpu().numpy()
        assert ids[:, :6].max() < cfg.vocab_size


def test_f32_engine_variants_agree_at_f32_round_off():
    """Round 3 moved the f32 decode loop onto the split residual form (f32 rows + per-16-column sums of squares), folded
    the cross-attention q-projection AND every layer's q / k / v projection into the neighbouring launches (layer 0:
    two table rows).  Every variant is the same function with different summation orders: against the r2 path
    (options = SINGLE_RESIDUAL_STREAM | SEPARATE_PROJECTIONS) teacher-forced logits at 96 positions
    agree to 2e-5 rel-L2 per (step, row), and each variant stays inside 1e-4 of the f32 oracle."""
    cfg = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg, seed=0, norm_scale_jitter=0.2)
    B, S = 5, 96
    x = _inputs(B, seed=31)
    x[4, 50:] = 0.0
    forced = _forced(B, S, 3)
    _, ref = _teacher_forced_ref(_oracle(cfg, params), x, forced)
    outs = {}
    if True:
        for name, opt in (("r3", 0), ("q-fold only", _lib.OPT_SEPARATE_QKV_PROJECTION),
                          ("separate projections", _lib.OPT_SEPARATE_PROJECTIONS),
                          ("r2 path", _lib.OPT_SINGLE_RESIDUAL_STREAM | _lib.OPT_SEPARATE_PROJECTIONS)):
            eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, options=opt)
            eng.load_params(params)
            assert eng.status(_lib.STATUS_Q_FOLD) == (0 if opt & (_lib.OPT_SEPARATE_PROJECTIONS | _lib.OPT_SINGLE_RESIDUAL_STREAM) else 1)
            assert eng.status(_lib.STATUS_QKV_FOLD) == (1 if opt == 0 else 0)
            assert eng.status(_lib.STATUS_RESIDUAL_SPLIT) == (0 if opt & _lib.OPT_SINGLE_RESIDUAL_STREAM else 1)
            eng.encode(torch.from_numpy(x).cuda())
            ids, logits = eng.decode_forced(forced, num_steps=S)
            outs[name] = logits.cpu().numpy()
            r = _rel_rows(outs[name], ref)
            print(f"f32 engine [{name}]: teacher-forced logits vs f32 oracle, {S} positions: max {r.max():.3e}")
            assert r.max() < 1e-4, (name, r.max())
            g = eng.decode(num_steps=24).cpu().numpy()
            outs[name + " ids"] = g
            del eng
    for name in ("r3", "q-fold only", "separate projections"):
        d = _rel_rows(outs[name], outs["r2 path"])
        print(f"f32 engine [{name}] vs the r2 path: max rel-L2 {d.max():.3e}")
        assert d.max() < 2e-5, (name, d.max())
        assert np.array_equal(outs[name + " ids"], outs["r2 path ids"])


def test_folded_qkv_projection_bf16_and_fp8_caches_against_the_separate_launches():
    """bf16 engine (the benched path) and bf16 + e4m3 caches: the q / k / v (+ cross-query) projection of every decoder
    layer rides in the previous layer's MLP out-projection launch (y_in . W = y2 . W + h . (Wo_mlp . W): a two-source
    K = mlp + emb product), layer 0's row is the sum of two table rows (embedding . W, position table . W), and the
    self-attention kernel applies 1/rms and rounds q / k / v itself.  Same function, rounded in different places:
    teacher-forced logits at 200 positions (incl. a padded row and a short segment) stay inside the bf16 bound against
    the f32 oracle for every variant, variants agree with each other to bf16 noise, graph replay == direct launches, 1
    chain == 3 chains, beam-1 and greedy both run."""
    cfg32 = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg32, seed=0, norm_scale_jitter=0.2)
    B, S = 7, 200
    x = _inputs(B, seed=41)
    x[6, 90:] = 0.0
    forced = _forced(B, S, 13)
    forced[2, 60:] = 0
    _, ref = _teacher_forced_ref(_oracle(cfg32, params), x, forced)
    for kv, bound in (("", 3e-2), ("fp8_e4m3", 6e-2)):
        cfg = dataclasses.replace(network.T5Config(), dtype="bfloat16", kv_dtype=kv)
        outs = {}
        for name, opt in (("folded", 0), ("q-fold only", _lib.OPT_SEPARATE_QKV_PROJECTION),
                          ("separate", _lib.OPT_SEPARATE_PROJECTIONS)):
            eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, options=opt)
            eng.load_params(params)
            assert eng.status(_lib.STATUS_QKV_FOLD) == (0 if opt & (_lib.OPT_SEPARATE_QKV_PROJECTION | _lib.OPT_SEPARATE_PROJECTIONS) else 1)
            eng.encode(torch.from_numpy(x).cuda())
            _, logits = eng.decode_forced(forced, num_steps=S)
            outs[name] = logits.cpu().numpy()
            r = _rel_rows(outs[name], ref)
            print(f"kv {kv or 'bf16'} [{name}]: teacher-forced logits vs f32 oracle max {r.max():.3e} mean {r.mean():.3e}")
            assert r.max() < bound, (kv, name, r.max())
            if not opt & (_lib.OPT_SEPARATE_QKV_PROJECTION | _lib.OPT_SEPARATE_PROJECTIONS):
                _, l2 = eng.decode_forced(forced, num_steps=40, use_graph=False)
                assert torch.equal(l2.cpu(), torch.from_numpy(outs[name][:40]))
                a = eng.decode(num_steps=48, chains=1).cpu().numpy()
                b = eng.decode(num_steps=48, chains=3).cpu().numpy()
                c = eng.decode(num_steps=48, use_graph=False).cpu().numpy()
                assert np.array_equal(a, b) and np.array_equal(a, c)
                d1 = eng.decode(num_steps=32, beam1=True, chains=1).cpu().numpy()
                d2 = eng.decode(num_steps=32, beam1=True, chains=2).cpu().numpy()
                assert np.array_equal(d1, d2)
                assert eng.status(_lib.STATUS_GRAPH_FALLBACKS) == 0
            del eng
        for name in ("folded", "q-fold only"):
            d = _rel_rows(outs[name], outs["separate"])
            print(f"kv {kv or 'bf16'} [{name}] vs separate launches: max {d.max():.3e} median {np.median(d):.3e}")
            assert d.max() < (2e-2 if not kv else 5e-2) and np.median(d) < 8e-3, (kv, name, d.max(), np.median(d))


def test_row_group_decode_schedule_is_bit_identical():
    """Batches of >= 128 rows decode as row groups on streams with hardware queues of their own, one host thread each
    (include/mt3_hip.h, "Schedule").  Rows are independent, so the ids must equal the single-stream graph-replayed schedule bit for bit:
    greedy, beam-1, with early exit (every group stops on its own rows, finished rows are retired), odd batch sizes;
    graph replay per group and direct launches."""
    cfg = network.T5Config(dtype="bfloat16", num_encoder_layers=2, num_decoder_layers=3)
    params = network.init_random_params(cfg, seed=5, norm_scale_jitter=0.1)
    k = params["decoder/logits_dense/kernel"].copy()
    k[:, 1] *= 3.0                                           # rows emit EOS at different steps
    params["decoder/logits_dense/kernel"] = k
    B = 131
    from mt3_amd import spectrograms, synthetic
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=8), None)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B)
    eng.load_params(params)
    eng.encode(lm)
    for kw in (dict(), dict(beam1=True)):
        a = eng.decode(num_steps=96, single_stream=True, **kw)
        assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 1 and eng.status(_lib.STATUS_LAST_DECODE_USED_GRAPH) == 1
        b = eng.decode(num_steps=96, **kw)
        assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 2, "a batch of 131 rows should run as two row groups"
        assert eng.status(_lib.STATUS_PARTITION_FALLBACKS) == 0
        assert torch.equal(a, b), kw
    full = eng.decode(num_steps=L, single_stream=True)
    ee = eng.decode(num_steps=L, early_exit=True)
    assert eng.steps_run <= L and torch.equal(ee, full)
    assert bool((full == 1).any()), "the case should contain rows that emit EOS"
    direct = eng.decode(num_steps=96, use_graph=False)
    assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 2 and eng.status(_lib.STATUS_LAST_DECODE_USED_GRAPH) == 0
    assert torch.equal(direct[:, :96], full[:, :96])
    # a small batch stays on the caller's stream, and the option switches the schedule off for good
    eng.encode(lm[:64])
    eng.decode(num_steps=8)
    assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 1
    e2 = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, options=_lib.OPT_NO_ROW_GROUPS)
    e2.load_params(params)
    e2.encode(lm)
    c = e2.decode(num_steps=96)
    assert e2.status(_lib.STATUS_LAST_DECODE_GROUPS) == 1 and torch.equal(c[:, :96], full[:, :96])


@pytest.mark.parametrize("dtype,B,groups,options", [
    ("float32", 259, 4, 0), ("float32", 130, 2, 0), ("bfloat16", 515, 4, 0),
    ("float32", 257, 4, _lib.OPT_SINGLE_RESIDUAL_STREAM | _lib.OPT_SEPARATE_PROJECTIONS)])
def test_row_group_counts_follow_operand_type_and_batch(dtype, B, groups, options):
    """The group count of the schedule (engine.hip: row_groups_for -- bf16 operands: 2 groups from 128 rows, 4 from
    512; f32: 2 from 128, 4 from 256): the ids of every count equal the single-stream schedule bit for bit (greedy,
    beam-1, early exit), for ragged group sizes and for the f32 engine's round-2 layout as well."""
    cfg = network.T5Config(dtype=dtype, num_encoder_layers=2, num_decoder_layers=3)
    params = network.init_random_params(cfg, seed=6, norm_scale_jitter=0.1)
    k = params["decoder/logits_dense/kernel"].copy()
    k[:, 1] *= 3.0
    params["decoder/logits_dense/kernel"] = k
    from mt3_amd import spectrograms, synthetic
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=9), None)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B, options=options)
    eng.load_params(params)
    eng.encode(lm)
    for kw in (dict(), dict(beam1=True)):
        a = eng.decode(num_steps=64, single_stream=True, **kw)
        assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 1
        b = eng.decode(num_steps=64, **kw)
        assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == groups and eng.status(_lib.STATUS_PARTITION_FALLBACKS) == 0
        assert torch.equal(a, b), kw
    full = eng.decode(num_steps=L, single_stream=True)
    ee = eng.decode(num_steps=L, early_exit=True)
    # under early exit f32 follows the bf16 rule since round 5 (two groups below 512 rows: the ragged regime is latency)
    ee_groups = groups if dtype == "bfloat16" or B < 256 else 2
    assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == ee_groups and torch.equal(ee, full)


def test_bench_batch_256_bf16_against_the_f32_engine_at_all_1024_positions():
    """the bench batch itself (B = 256) was covered by self-consistency only -- the CPU oracle
    stops at B = 64.  The f32 ENGINE is token-exact against the oracle (32 x 256 greedy steps) and within 1.3e-6 of it
    teacher-forced, so it stands in for the oracle here: both engines are teacher-forced with the f32 engine's own
    greedy stream over ALL 1024 cache positions of 256 synthetic segments (262,144 (step, row) pairs, compared on the
    GPU).  bf16 (the benched dtype): rel-L2 < 3e-2 at every pair, no drift with depth, arg-max equal wherever the f32
    top-2 margin exceeds 0.05 sigma; the overall arg-max agreement is printed."""
    from mt3_amd import spectrograms, synthetic
    B = 256
    cfg32 = network.T5Config(dtype="float32")
    params = network.init_random_params(cfg32, seed=0)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=1000), None)
    e32 = network.Transformer(cfg32, input_length=256, max_decode_length=L, max_batch=B)
    e32.load_params(params)
    e32.encode(lm)
    stream = e32.decode(num_steps=L)                                      # the f32 engine's free-running greedy ids
    _, ref = e32.decode_forced(stream)                                    # [L, B, V] f32 on the GPU
    del e32
    e16 = network.Transformer(network.T5Config(dtype="bfloat16"), input_length=256, max_decode_length=L, max_batch=B)
    e16.load_params(params)
    e16.encode(lm)
    _, got = e16.decode_forced(stream)
    assert e16.status(_lib.STATUS_QKV_FOLD) == 1
    r = torch.empty(L, B, device="cuda", dtype=torch.float64)
    agree = torch.empty(L, B, device="cuda", dtype=torch.bool)
    safe = torch.empty(L, B, device="cuda", dtype=torch.bool)
    for s0 in range(0, L, 64):                                            # chunks: keeps the f64 temporaries small
        a, b = got[s0:s0 + 64].double(), ref[s0:s0 + 64].double()
        r[s0:s0 + 64] = (a - b).norm(dim=-1) / b.norm(dim=-1)
        top2 = b.topk(2, dim=-1).values
        safe[s0:s0 + 64] = (top2[..., 0] - top2[..., 1]) > 0.05 * b.std(-1)
        agree[s0:s0 + 64] = a.argmax(-1) == b.argmax(-1)
    print(f"B = 256 bf16 vs the f32 engine, teacher-forced on the f32 stream, 256 x 1024 positions: rel-L2 max "
          f"{float(r.max()):.3e} mean {float(r.mean()):.3e}; first / last 64 positions {float(r[:64].mean()):.3e} / "
          f"{float(r[-64:].mean()):.3e}; arg-max agreement {float(agree.double().mean()):.4f} overall, "
          f"{float(safe.double().mean()):.3f} of the positions have a margin > 0.05 sigma")
    assert float(r.max()) < 3e-2
    assert float(r[-64:].mean()) < 1.5 * float(r[:64].mean()) + 1e-3
    assert bool(agree[safe].all()) and float(safe.double().mean()) > 0.5
    assert float(agree.double().mean()) > 0.97
