"""The DECODE step of a large f32 engine on the bf16 pipes (round 5; mt3_engine::dec_x6, include/mt3_hip.h
MT3_OPT_DECODE_F32_MFMA).  An f32 engine of >= 512 slots multiplies the step's dense layers (network.Decoder's DenseGeneral
calls, mt3/network.py:88-155, mt3/layers.py:373-418) with every f32 operand as three exact bf16 planes -- the tiles its
encoder already uses -- because at >= 128 rows per row group those launches are compute-bound, not latency-bound
(profiles/r5_refill_f32_kernel_stats.csv).  Not a reduced-precision mode (tests/test_three_plane_arithmetic.py): here the
engine is held against the f32 oracle at the f32 bounds -- teacher-forced logits at every one of 96 cache positions,
greedy tokens exact -- and against the same engine on the f32 matrix instruction (option bit) to f32 round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402
from oracle import network as ON  # noqa: E402


def test_large_f32_engine_decodes_on_three_planes_and_matches_the_oracle():
    cfg = network.T5Config(dtype="float32", num_encoder_layers=2, num_decoder_layers=3)
    params = network.init_random_params(cfg, seed=9, norm_scale_jitter=0.2)
    B, S = 12, 96
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(B, seed=14), None)
    lm[5, 80:] = 0.0                                              # a short segment
    orc = ON.Oracle(params, ON.T5Config(num_encoder_layers=2, num_decoder_layers=3))
    torch.set_num_threads(16)
    with torch.no_grad():
        enc_ref = orc.encode(lm.cpu().numpy())
        ids_ref, logits_ref = orc.greedy_decode(enc_ref, S, return_logits=True)
    logits_ref = logits_ref.numpy()                               # [B, S, V]
    got = {}
    for name, opt in (("three bf16 planes", 0), ("f32 instruction", _lib.OPT_DECODE_F32_MFMA)):
        eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=512, options=opt)
        eng.load_params(params)
        assert eng.status(_lib.STATUS_QKV_FOLD) == (0 if opt == 0 else 1)      # the large engine runs without the folds
        eng.encode(lm)
        ids = eng.decode(num_steps=S).cpu().numpy()
        forced, logits = eng.decode_forced(ids_ref, num_steps=S)               # the oracle's own tokens as inputs
        logits = logits.cpu().numpy().transpose(1, 0, 2)                       # [B, S, V]
        rel = np.linalg.norm(logits - logits_ref, axis=-1) / np.linalg.norm(logits_ref, axis=-1)
        print(f"large f32 engine [{name}]: teacher-forced logits vs the f32 oracle, max rel-L2 over {B} x {S}: {rel.max():.3e}")
        assert rel.max() < 1e-4, (name, float(rel.max()))
        assert np.array_equal(ids[:, :S], ids_ref), (name, "greedy tokens differ from the oracle's")
        got[name] = logits
        del eng
    d = np.linalg.norm(got["three bf16 planes"] - got["f32 instruction"], axis=-1) / np.linalg.norm(got["f32 instruction"], axis=-1)
    print(f"large f32 engine: three planes vs the f32 instruction, max rel-L2 {d.max():.3e}")
    assert d.max() < 1e-5


def test_large_f32_engine_full_width_row_groups_retirement_and_refill_agree():
    """512 slots in use: four row groups of 128 rows on the three-plane tiles; the every-row schedule, early exit with
    retirement (the groups shrink below one 128-row tile: the same kernels, the same bits) and in-flight batching return
    the same ids."""
    cfg = network.T5Config(dtype="float32", num_encoder_layers=1, num_decoder_layers=2)
    params = network.init_random_params(cfg, seed=10, norm_scale_jitter=0.1)
    k = params["decoder/logits_dense/kernel"].copy()
    k[:, 1] *= 3.0
    params["decoder/logits_dense/kernel"] = k
    B, N, S = 512, 700, 96
    eng = network.Transformer(cfg, input_length=256, max_decode_length=1024, max_batch=B)
    eng.load_params(params)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(N, seed=15), None)
    lens = np.clip(np.rint(np.random.default_rng(3).normal(40, 20, N)), 1, S + 20).astype(np.int32)
    try:
        ref = []
        for a in range(0, N, B):
            eng.encode(lm[a:a + B])
            eng.debug_set_eos_schedule(lens[a:a + B])
            full = eng.decode(num_steps=S, single_stream=True)
            if a == 0:
                assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 1
                groups = eng.decode(num_steps=S)
                assert eng.status(_lib.STATUS_LAST_DECODE_GROUPS) == 4 and torch.equal(groups, full)
                early = eng.decode(num_steps=S, early_exit=True)
                assert torch.equal(early, full) and eng.status(_lib.STATUS_LAST_DECODE_COMPACTIONS) >= 1
            ref.append(full)
        ref = torch.cat(ref)
        eng.debug_set_eos_schedule(lens)
        got = eng.transcribe(lm, num_steps=S)
        assert torch.equal(got, ref), (got != ref).any(1).nonzero().flatten().tolist()[:8]
    finally:
        eng.debug_set_eos_schedule(None)
