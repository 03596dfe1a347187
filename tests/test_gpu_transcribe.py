"""In-flight batching: mt3_engine_transcribe (round 5; VERDICT r4 "next round" #1), through the C ABI.

The reference transcribes a list of segments as a loop over `.batch(8)` calls of t5x predict_batch_with_aux (NB:295-301,
mt3/models.py:121-152), and every call runs until its LAST row has terminated (t5x decoding.beam_search).  The engine's
streaming entry keeps `max_batch` decode slots busy instead: a slot whose segment has finished hands its id row over and
restarts at position 0 on the next encoded segment (per-slot position counter, as the reference's cache index is per call:
mt3/layers.py:246-314).  Segments are independent, so row i of the result must be BIT-IDENTICAL to what a plain
encode + decode of segment i returns -- whatever slot it ran in, whatever ran in that slot before it, whatever the caches
held: greedy and beam-1, f32 / bf16 / e4m3 caches, row groups and one stream, graph replay and direct launches, rows that
emit EOS of their own accord mixed with imposed lengths (the synthetic EOS schedule of include/mt3_hip_debug.h, here PER
SEGMENT), rows that never finish, NaN-poisoned caches.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network, spectrograms, synthetic  # noqa: E402

L = 1024


def _engine(dtype, B, kv="", eos_boost=3.0, dec_layers=3, seed=5, dense=""):
    cfg = network.T5Config(dtype=dtype, kv_dtype=kv, dense_dtype=dense, num_encoder_layers=2, num_decoder_layers=dec_layers)
    params = network.init_random_params(cfg, seed=seed, norm_scale_jitter=0.1)
    if eos_boost:
        k = params["decoder/logits_dense/kernel"].copy()
        k[:, 1] *= eos_boost                                 # some rows emit EOS of their own accord, at different steps
        params["decoder/logits_dense/kernel"] = k
    eng = network.Transformer(cfg, input_length=256, max_decode_length=L, max_batch=B)
    eng.load_params(params)
    return eng


def _lengths(n, mean, sd, hi, seed):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(mean, sd, n)), 1, hi).astype(np.int32)


def _reference(eng, lm, steps, lens, beam1, chunk):
    """plain encode + decode (every row, every step; one stream) of the segments in calls of `chunk`.  A last call of
    fewer than 8 segments is moved back to cover 8 (its first rows are dropped again): encoder passes of fewer than
    2048 rows take the decode-sized GEMM tiles, whose sums are rounded in other places (include/mt3_hip.h,
    mt3_engine_encode) -- the streaming entry never issues such a pass unless the whole job is that small."""
    N, out = lm.shape[0], []
    for a in range(0, N, chunk):
        b = min(a + chunk, N)
        a0 = max(0, b - 8) if b - a < 8 else a
        eng.encode(lm[a0:b])
        if lens is not None:
            eng.debug_set_eos_schedule(lens[a0:b])
        out.append(eng.decode(num_steps=steps, single_stream=True, beam1=beam1)[a - a0:])
    return torch.cat(out, 0)


@pytest.mark.parametrize("dtype,kv,B,groups", [("float32", "", 136, 2), ("bfloat16", "", 136, 2), ("float32", "", 260, 4),
                                               ("bfloat16", "fp8_e4m3", 140, 2)])
def test_refilled_slots_decode_every_segment_bit_identically(dtype, kv, B, groups):
    S = 160
    N = 3 * B + 17                                            # the last reference call is ragged; 2 B + 17 segments wait
    eng = _engine(dtype, B, kv)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(N, seed=21), None)
    lens = _lengths(N, 60, 30, S + 30, seed=N)               # a few segments outlive the S steps: they end at the cap
    lens[:4] = (1, 2, S, S + 50)
    lens[B:B + 3] = (1, S, 2)                                # ... also among the segments that arrive by refill
    try:
        for beam1 in (False, True):
            ref = _reference(eng, lm, S, lens, beam1, B)
            r = ref.cpu().numpy()
            assert (r[:, :S] == 1).any(1).mean() > 0.8
            eng.debug_set_eos_schedule(lens)                 # one schedule entry per SEGMENT
            # (the last one: mt3_debug_engine_transcribe -- another poll interval, another number of row groups)
            for kw in (dict(), dict(use_graph=False), dict(single_stream=True), dict(debug_poll_steps=8, debug_row_groups=3)):
                got = eng.transcribe(lm, num_steps=S, beam1=beam1, **kw)
                st = eng.transcribe_stats
                bad = (got != ref).any(1).nonzero().flatten().tolist()
                assert not bad, (dtype, kv, beam1, kw, bad[:8], st)
                assert st["slots"] == B and st["refills"] == N - B and st["encoder_chunks"] >= 1, st
                assert st["groups"] == (1 if kw.get("single_stream") else kw.get("debug_row_groups") or groups), st
                assert st["used_graph"] == (0 if kw.get("use_graph") is False else 1), st
                assert eng.status(_lib.STATUS_GRAPH_FALLBACKS) == 0
            # far fewer steps than one batch-synchronous call after the other would take
            assert st["steps_run"] < 0.8 * S * -(-N // B), st
        # stale cache contents (NaN in every cache format, cross-attention caches included): the same ids
        eng.debug_poison_caches(0xFF, cross=True)
        again = eng.transcribe(lm, num_steps=S, beam1=True)
        assert torch.equal(again, ref)
    finally:
        eng.debug_set_eos_schedule(None)
    # the engine is an ordinary engine afterwards
    eng.encode(lm[:B])
    a = eng.decode(num_steps=64, single_stream=True)
    b = eng.decode(num_steps=64, early_exit=True)
    assert torch.equal(a, b)


def test_natural_eos_only_and_small_engines():
    """No imposed lengths: rows that emit EOS of their own accord (boosted EOS column) and rows that never do.
    (a) 40 slots, one row group, 117 segments; (b) an 8-slot engine -- the reference InferenceModel's batch size
    (NB:190) -- whose staging chunks are 8 segments, the last one padded with segments already handed out;
    (c) fewer segments than slots: no refill at all."""
    for dtype, B, N, S in (("float32", 40, 117, 256), ("bfloat16", 8, 29, 200), ("float32", 16, 5, 128)):
        eng = _engine(dtype, B, eos_boost=4.0, seed=11)
        lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(N, seed=3), None)
        for beam1 in (False, True):
            ref = _reference(eng, lm, S, None, beam1, max(B, 8))
            got = eng.transcribe(lm, num_steps=S, beam1=beam1)
            assert torch.equal(got, ref), (dtype, B, N, beam1, (got != ref).any(1).nonzero().flatten().tolist()[:8])
            st = eng.transcribe_stats
            assert st["slots"] == min(B, N) and st["refills"] == max(0, N - B) and st["groups"] == 1, st
            assert bool((ref == 1).any()), "the case should contain rows that emit EOS"


def test_mxfp8_encoder_and_e4m3_caches_through_the_staging_ring():
    """BASELINE configs[4]'s ingredients together: the MXFP8 encoder writes a chunk's cross-K/V as bf16 into the landing
    buffer, the quantiser turns them into e4m3 rows + power-of-two scales IN THE STAGING CHUNK, and the refill copies rows
    and scales into the slot's caches.  Same ids as plain calls of the same engine."""
    B, N, S = 24, 100, 96
    eng = _engine("bfloat16", B, kv="fp8_e4m3", dense="fp8_e4m3", dec_layers=2)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(N, seed=41), None)
    lens = _lengths(N, 30, 12, S, seed=9)
    try:
        ref = _reference(eng, lm, S, lens, False, B)
        eng.debug_set_eos_schedule(lens)
        got = eng.transcribe(lm, num_steps=S)
    finally:
        eng.debug_set_eos_schedule(None)
    assert torch.equal(got, ref), ((got != ref).any(1).nonzero().flatten().tolist()[:8], eng.transcribe_stats)
    assert eng.status(_lib.STATUS_DENSE_FP8) == 1 and eng.status(_lib.STATUS_KV_FP8) == 1


def test_a_long_queue_through_few_slots_keeps_the_ring_turning():
    """More segments than the staging ring holds (8 chunks): chunks are reused many times, consumers give them back at the
    poll after they took them.  24 slots, chunks of 24, 700 short segments."""
    B, N, S = 24, 700, 96
    eng = _engine("bfloat16", B, dec_layers=2)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(N, seed=33), None)
    lens = _lengths(N, 24, 10, S, seed=7)
    big = _engine("bfloat16", 100, dec_layers=2)             # the reference: plain calls of 100 segments (same weights)
    ref = _reference(big, lm, S, lens, False, 100)
    del big
    eng.debug_set_eos_schedule(lens)
    try:
        got = eng.transcribe(lm, num_steps=S)
    finally:
        eng.debug_set_eos_schedule(None)
    st = eng.transcribe_stats
    assert torch.equal(got, ref), ((got != ref).any(1).nonzero().flatten().tolist()[:8], st)
    assert st["encoder_chunks"] == -(-(N - B) // B) and st["encoder_chunks"] > 8, st


def test_transcribe_argument_errors():
    eng = _engine("bfloat16", 8, dec_layers=1, eos_boost=0.0)
    lm = spectrograms.compute_spectrogram_batch(synthetic.synth_audio(9, seed=1), None)
    lib = _lib.load()
    ids = torch.empty((9, L), device="cuda", dtype=torch.int32)
    s = torch.cuda.current_stream().cuda_stream
    assert lib.mt3_engine_transcribe(eng._h, lm.data_ptr(), 9, L, _lib.DECODE_ASYNC, ids.data_ptr(), None, s) == _lib.MT3_ERR_INVALID
    assert lib.mt3_engine_transcribe(eng._h, lm.data_ptr(), 9, L + 1, 0, ids.data_ptr(), None, s) == _lib.MT3_ERR_INVALID
    assert lib.mt3_engine_transcribe(eng._h, None, 9, L, 0, ids.data_ptr(), None, s) == _lib.MT3_ERR_INVALID
    assert lib.mt3_engine_transcribe(eng._h, lm.data_ptr(), 0, L, 0, ids.data_ptr(), None, s) == _lib.MT3_ERR_INVALID
    eng.debug_set_eos_schedule(np.full(4, 5, np.int32))      # a schedule for 4 of the 9 segments: entries 4 .. 7 = never
    try:
        got = eng.transcribe(lm[:8], num_steps=32)
        assert (got[:4, :5] == 1).any(1).all() and not got[:4, 5:].any()
        with pytest.raises(_lib.Mt3Error):                   # ... but the 9th segment would index past the schedule
            eng.transcribe(lm, num_steps=32)
        eng.debug_set_eos_schedule(np.full(9, 5, np.int32))  # the schedule grows with the corpus
        got = eng.transcribe(lm, num_steps=32)
        assert (got[:, :5] == 1).any(1).all() and not got[:, 5:].any()
    finally:
        eng.debug_set_eos_schedule(None)
