"""The f32 engine's encoder on the bf16 pipes (round 4; gemm.hip: gemm_x6_kernel).  Every f32 operand is split EXACTLY into
three bf16 terms and the six significant products accumulate in f32: not a reduced-precision mode -- the instruction-level
check (tools/micro/mfma_bf16_accuracy.hip) puts it at 1.3e-7 of sum |p| where the f32 matrix instruction is at 2.1e-7.
Here: the engine's encoder output and first-step logits against the f32 oracle at the SAME bounds as before (1e-4), and
against the engine that keeps the f32 instruction (option bit) to f32 round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network  # noqa: E402
from oracle import frontend as OF  # noqa: E402
from oracle import network as ON  # noqa: E402


@pytest.mark.parametrize("shape", ["mt3", "base_2_layers"])
def test_f32_encoder_on_three_bf16_planes_is_as_exact_as_the_f32_instruction(shape):
    import dataclasses
    if shape == "mt3":
        cfg = network.T5Config(dtype="float32")
    else:     # emb 768, 12 heads, mlp 2048: N = 2304 / 768 / 4096 tiles, K = 768 / 2048
        cfg = dataclasses.replace(network.MT3_BASE, dtype="float32", num_encoder_layers=2, num_decoder_layers=2)
    params = network.init_random_params(cfg, seed=3, norm_scale_jitter=0.2)
    B = 9                                                      # 2304 rows: the encoder-sized tile, a ragged last tile? no: 18 tiles
    audio = OF.synth_audio(B, seed=12)
    x = np.stack([OF.compute_logmel(a, np.float32) for a in audio])
    x[4, 100:] = 0.0                                           # a short segment (zero rows after the log)
    torch.set_num_threads(16)
    orc = ON.Oracle(params, ON.T5Config(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_heads=cfg.num_heads,
                                        num_encoder_layers=cfg.num_encoder_layers,
                                        num_decoder_layers=cfg.num_decoder_layers, mlp_dim=cfg.mlp_dim))
    decode = shape == "mt3"          # (the f32 DECODE loop of the emb-768 shape is not built: its tiles hold K <= 512 partial sums)
    with torch.no_grad():
        enc_ref = orc.encode(x)
        logits_ref = orc.decode_logits(enc_ref, np.zeros((B, 1), np.int32))[:, 0].numpy() if decode else None
    enc_ref = enc_ref.numpy()
    outs = {}
    for name, opt in (("three bf16 planes", 0), ("f32 instruction", _lib.OPT_ENCODER_F32_MFMA)):
        eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B, options=opt)
        eng.load_params(params)
        enc = eng.encode(torch.from_numpy(x).cuda(), return_encoded=True).cpu().numpy()
        rel = [float(np.linalg.norm(enc[b] - enc_ref[b]) / np.linalg.norm(enc_ref[b])) for b in range(B)]
        outs[name] = [enc]
        rl = 0.0
        if decode:
            ids, logits0 = eng.decode(num_steps=24, return_first_logits=True)
            outs[name] += [logits0.cpu().numpy(), ids.cpu().numpy()]
            rl = float(np.linalg.norm(outs[name][1] - logits_ref) / np.linalg.norm(logits_ref))
        print(f"f32 encoder [{shape}, {name}]: rel-L2 vs the f32 oracle max {max(rel):.3e}; step-0 logits {rl:.3e}")
        assert max(rel) < 1e-4 and rl < 1e-4, (name, max(rel), rl)
        del eng
    a, b = outs["three bf16 planes"], outs["f32 instruction"]
    d = max(float(np.linalg.norm(a[0][i] - b[0][i]) / np.linalg.norm(b[0][i])) for i in range(B))
    print(f"f32 encoder [{shape}]: three bf16 planes vs the f32 instruction: max rel-L2 {d:.3e}")
    assert d < 5e-6, d
    if decode:
        assert np.array_equal(a[2], b[2]), "greedy ids of the two evaluations differ"


def test_f32_encoder_output_does_not_depend_on_the_batch_a_segment_sits_in():
    """The same segment encoded inside 256 and inside 40 segments (different grids, different XCD dealing of the tiles):
    bit-identical rows."""
    cfg = network.T5Config(dtype="float32", num_encoder_layers=2, num_decoder_layers=1)
    params = network.init_random_params(cfg, seed=4, norm_scale_jitter=0.2)
    from mt3_amd import spectrograms, synthetic
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(256, seed=6), None)
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=256)
    eng.load_params(params)
    big = eng.encode(lm, return_encoded=True).clone()
    small = eng.encode(lm[:40], return_encoded=True).clone()
    assert torch.equal(big[:40], small)


def test_f32_encoder_on_three_planes_at_t512():
    """ismir2021 preset (T = 512 encoder frames, NB:176-179) at a batch that takes the encoder-sized tile (B x 512 >= 2048
    rows): the three-plane dense layers with the 512-row sinusoid table (POS epilogue) and the head-major cross-K/V of
    512 keys (HEADS epilogue), next to enc_attn_split_kernel<float, 512>: encoder and step-0 logits against the f32 oracle
    at the f32 bounds, and against the engine that keeps the f32 instruction."""
    import dataclasses
    cfg = dataclasses.replace(network.T5Config(dtype="float32"), vocab_size=1664, num_encoder_layers=3,
                              num_decoder_layers=2)
    params = network.init_random_params(cfg, seed=5, norm_scale_jitter=0.2)
    B = 5                                                       # 2560 rows = 20 tiles
    audio = OF.synth_audio(B * 2, seed=21).reshape(B, -1)
    x = np.stack([OF.compute_logmel(a, np.float32) for a in audio])
    assert x.shape[1] == 512
    x[1, 333:] = 0.0
    torch.set_num_threads(16)
    orc = ON.Oracle(params, ON.T5Config(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_heads=cfg.num_heads,
                                        num_encoder_layers=cfg.num_encoder_layers,
                                        num_decoder_layers=cfg.num_decoder_layers, mlp_dim=cfg.mlp_dim))
    with torch.no_grad():
        enc_ref = orc.encode(x)
        logits_ref = orc.decode_logits(enc_ref, np.zeros((B, 1), np.int32))[:, 0].numpy()
    enc_ref = enc_ref.numpy()
    outs = {}
    for name, opt in (("three bf16 planes", 0), ("f32 instruction", _lib.OPT_ENCODER_F32_MFMA)):
        eng = network.Transformer(cfg, input_length=512, max_decode_length=1024, max_batch=B, options=opt)
        eng.load_params(params)
        enc = eng.encode(torch.from_numpy(x).cuda(), return_encoded=True).cpu().numpy()
        rel = [float(np.linalg.norm(enc[b] - enc_ref[b]) / np.linalg.norm(enc_ref[b])) for b in range(B)]
        ids, logits0 = eng.decode(num_steps=16, return_first_logits=True)
        rl = float(np.linalg.norm(logits0.cpu().numpy() - logits_ref) / np.linalg.norm(logits_ref))
        print(f"f32 encoder, T = 512 [{name}]: rel-L2 vs the f32 oracle max {max(rel):.3e}; step-0 logits {rl:.3e}")
        assert max(rel) < 1e-4 and rl < 1e-4, (name, max(rel), rl)
        outs[name] = (enc, ids.cpu().numpy())
        del eng
    a, b = outs["three bf16 planes"], outs["f32 instruction"]
    d = max(float(np.linalg.norm(a[0][i] - b[0][i]) / np.linalg.norm(b[0][i])) for i in range(B))
    print(f"f32 encoder, T = 512: three bf16 planes vs the f32 instruction: max rel-L2 {d:.3e}")
    assert d < 5e-6, d
    assert np.array_equal(a[1], b[1]), "greedy ids of the two evaluations differ"
