"""GPU parity tests of the whole engine (encode + cached greedy decode) against the CPU oracle,
through the C ABI.  Random-init weights (reference initialisers), synthetic audio.

Tolerances (confirmed against the measured f32 noise floor):
  f32 path : encoder output / step-0 logits rel-L2 < 1e-4 and max-abs < 2e-4 * max|ref|;
             greedy token stream identical to the oracle's.
  bf16 path: encoder output rel-L2 < 2e-2 and cosine > 0.999; step-0 logits rel-L2 < 3e-2.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import _lib, network  # noqa: E402
from oracle import frontend as OF  # noqa: E402
from oracle import network as ON  # noqa: E402

T, L = 256, 1024


def _inputs(B, seed=0):
    audio = OF.synth_audio(B, seed=seed)
    return np.stack([OF.compute_logmel(a, np.float64).astype(np.float32) for a in audio])


def _params(cfg, seed=0, eos_boost=1.0):
    p = network.init_random_params(cfg, seed=seed, norm_scale_jitter=0.2)
    if eos_boost != 1.0:
        p["decoder/logits_dense/kernel"] = p["decoder/logits_dense/kernel"].copy()
        p["decoder/logits_dense/kernel"][:, 1] *= eos_boost
    return p


def _oracle(cfg, params):
    oc = ON.T5Config(vocab_size=cfg.vocab_size, emb_dim=cfg.emb_dim, num_heads=cfg.num_heads,
                     num_encoder_layers=cfg.num_encoder_layers, num_decoder_layers=cfg.num_decoder_layers,
                     head_dim=cfg.head_dim, mlp_dim=cfg.mlp_dim, input_depth=cfg.input_depth)
    return ON.Oracle(params, oc)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def setup():
    cfg32 = network.T5Config(dtype="float32")
    params = _params(cfg32, seed=0, eos_boost=2.5)
    x = _inputs(3, seed=0)
    x[2, 100:] = 0.0                                    # a short segment: zero rows after the log (F8)
    orc = _oracle(cfg32, params)
    enc_ref = orc.encode(x)
    ids_ref, logits_ref = orc.greedy_decode(enc_ref, 48, return_logits=True)
    return dict(params=params, x=x, enc_ref=enc_ref.numpy(), ids_ref=ids_ref, logits_ref=logits_ref.numpy())


def _engine(dtype, params, B, options=0):
    cfg = network.T5Config(dtype=dtype)
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=B, options=options)
    eng.load_params(params)
    return eng


def test_engine_f32_matches_oracle(setup):
    eng = _engine("float32", setup["params"], 3)
    enc = eng.encode(torch.from_numpy(setup["x"]).cuda(), return_encoded=True).cpu().numpy()
    r = rel(enc, setup["enc_ref"])
    assert r < 1e-4, f"encoder rel-L2 {r}"
    assert np.abs(enc - setup["enc_ref"]).max() < 2e-4 * np.abs(setup["enc_ref"]).max()
    ids, logits0 = eng.decode(num_steps=48, return_first_logits=True)
    ids, logits0 = ids.cpu().numpy(), logits0.cpu().numpy()
    r = rel(logits0, setup["logits_ref"][:, 0])
    assert r < 1e-4, f"step-0 logits rel-L2 {r}"
    ref = setup["ids_ref"]
    if not np.array_equal(ids[:, :48], ref):
        bad = np.argwhere(ids[:, :48] != ref)
        raise AssertionError(f"greedy tokens diverge first at (row, step) {bad[0]}; "
                             f"got {ids[bad[0][0], :48]} want {ref[bad[0][0]]}")
    assert np.all(ids[:, 48:] == 0)
    # EOS bookkeeping: after a row's first 1 every id is 0
    for row in ids:
        hit = np.nonzero(row == 1)[0]
        if hit.size:
            assert np.all(row[hit[0] + 1:] == 0)
    assert any((row == 1).any() for row in ref), "test weights should make at least one row emit EOS"


def test_engine_bf16_within_tolerance(setup):
    eng = _engine("bfloat16", setup["params"], 3)
    enc = eng.encode(torch.from_numpy(setup["x"]).cuda(), return_encoded=True).cpu().numpy()
    ref = setup["enc_ref"]
    for b in range(3):
        r = rel(enc[b], ref[b])
        cos = float((enc[b] * ref[b]).sum() / (np.linalg.norm(enc[b]) * np.linalg.norm(ref[b])))
        assert r < 2e-2 and cos > 0.999, f"segment {b}: rel-L2 {r}, cosine {cos}"
    ids, logits0 = eng.decode(num_steps=8, return_first_logits=True)
    r = rel(logits0.cpu().numpy(), setup["logits_ref"][:, 0])
    assert r < 3e-2, f"step-0 logits rel-L2 {r}"
    # argmax agrees wherever the oracle's top-1 / top-2 margin is comfortable
    lr = setup["logits_ref"][:, 0]
    top2 = np.sort(lr, -1)[:, -2:]
    safe = (top2[:, 1] - top2[:, 0]) > 0.05 * lr.std()
    got = logits0.cpu().numpy().argmax(-1)
    assert np.array_equal(got[safe], lr.argmax(-1)[safe])


def test_beam1_decode_matches_oracle():
    """MT3_DECODE_BEAM1 = the selection rule of t5x beam_search(num_decodes=1), against the
    oracle's emulation.  Flat logits with a boosted EOS column give rows that (a) finish on a top-1 EOS like
    greedy, (b) finish EARLIER than greedy through a top-2 EOS, (c) never finish and return the live row."""
    cfg32 = network.T5Config(dtype="float32")
    params = _params(cfg32, seed=1)
    k = params["decoder/logits_dense/kernel"].copy() * 0.3
    k[:, 1] *= 1.5
    params["decoder/logits_dense/kernel"] = k
    x = _inputs(6, seed=3)
    x[2, 100:] = 0.0
    orc = _oracle(cfg32, params)
    enc_ref = orc.encode(x)
    steps = 32
    greedy_ref, _ = orc.greedy_decode(enc_ref, steps, return_logits=True)
    ref = orc.beam1_decode(enc_ref, steps)
    assert (ref != greedy_ref).any(axis=1).sum() >= 2          # the case set does discriminate the two rules
    assert not (ref == 1).any(axis=1).all()                    # ... and holds a row that never finishes
    eng = _engine("float32", params, 6)
    eng.encode(torch.from_numpy(x).cuda())
    got = {}
    for name, kw in (("graph", {}), ("direct", dict(use_graph=False)),
                     ("early_exit", dict(early_exit=True))):
        ids = eng.decode(num_steps=steps, beam1=True, **kw).cpu().numpy()
        assert (ids[:, steps:] == 0).all()
        got[name] = ids[:, :steps]
    for name, ids in got.items():
        if name == "early_exit":        # rows that never finish keep the search alive: same result
            assert eng.steps_run <= steps
        assert np.array_equal(ids, ref), f"{name}: beam-1 tokens differ from the oracle\n{ids}\n{ref}"
    # greedy on the same engine still follows the greedy oracle
    ids = eng.decode(num_steps=steps).cpu().numpy()[:, :steps]
    assert np.array_equal(ids, greedy_ref)


@pytest.mark.parametrize("dtype", ["bfloat16", "float32"])
def test_graph_replay_equals_direct_launch(setup, dtype):
    eng = _engine(dtype, setup["params"], 3)
    x = torch.from_numpy(setup["x"]).cuda()
    eng.encode(x)
    a = eng.decode(num_steps=64, use_graph=True).cpu().numpy()
    eng.encode(x)
    b = eng.decode(num_steps=64, use_graph=False).cpu().numpy()
    assert np.array_equal(a, b), "hipGraph replay and direct launches must be bit-identical"
    # decode twice from the same encode state is reproducible (cache fully rewritten)
    c = eng.decode(num_steps=64, use_graph=True).cpu().numpy()
    assert np.array_equal(a, c)


def test_full_length_decode_and_early_exit(setup):
    eng = _engine("bfloat16", setup["params"], 3)
    x = torch.from_numpy(setup["x"]).cuda()
    eng.encode(x)
    full = eng.decode().cpu().numpy()                    # all 1024 steps
    assert eng.steps_run == L and full.shape == (3, L)
    eng.encode(x)
    early = eng.decode(early_exit=True).cpu().numpy()
    from mt3_amd import vocabularies as V
    vocab = V.GenericTokenVocabulary(1388, extra_ids=100)
    tf, te = vocab.decode_tf(full), vocab.decode_tf(early)
    for a, b in zip(tf, te):                             # identical up to and including EOS
        n = int(np.argmax(a == -1)) if (a == -1).any() else L
        if (a == -1).any() and eng.steps_run >= n + 1:
            assert np.array_equal(a[: n + 1], b[: n + 1])
    # smaller batch than max_batch, and batch change re-captures the graph
    eng.encode(x[:2])
    two = eng.decode(num_steps=16).cpu().numpy()
    assert np.array_equal(two[:, :16], full[:2, :16])


def test_decode_chains_are_bit_identical(setup):
    """The batch dealt to 1 / 2 / 4 / 8 parallel graph branches decodes to exactly the same ids."""
    B = 64
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, T, 512, device="cuda", generator=g) * 2 - 4
    cfg = network.T5Config(dtype="bfloat16")
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=B, decode_chains=1)
    eng.load_params(setup["params"])
    eng.encode(x)
    ref = eng.decode(num_steps=24, chains=1).cpu().numpy()
    for n in (2, 4, 8):
        for graph in (True, False):
            got = eng.decode(num_steps=24, chains=n, use_graph=graph).cpu().numpy()
            assert np.array_equal(got, ref), f"chains={n} graph={graph}"
    # ragged split: 50 rows over 3 chains
    eng.encode(x[:50])
    a = eng.decode(num_steps=8, chains=1).cpu().numpy()
    b = eng.decode(num_steps=8, chains=3).cpu().numpy()
    assert np.array_equal(a, b)
    # the beam-1 selection keeps its per-row state in the same row-sliced arrays
    a = eng.decode(num_steps=16, chains=1, beam1=True).cpu().numpy()
    b = eng.decode(num_steps=16, chains=3, beam1=True).cpu().numpy()
    assert np.array_equal(a, b)


def test_split_residual_stream_matches_the_single_f32_stream(setup):
    """bf16 decode keeps the residual rows as f32 + a bf16 copy + exact per-16-column sums of squares (DESIGN.md
    section 2); options = MT3_OPT_SINGLE_RESIDUAL_STREAM keeps the single f32 stream with in-kernel
    statistics.  Same MFMA operands, row scales equal up to the summation order: step-0 logits must agree to
    f32 round-off (a wrong partial sum would show as a percent-level shift that the bf16 tolerances could hide),
    and the greedy tokens must be identical."""
    x = torch.from_numpy(np.repeat(setup["x"], 12, axis=0)[:34]).cuda()          # ragged: 34 rows = 32 + 2
    out = {}
    # (the folded projections round differently: their own test below)
    for name, opt in (("split", _lib.OPT_SEPARATE_PROJECTIONS),
                      ("single", _lib.OPT_SEPARATE_PROJECTIONS | _lib.OPT_SINGLE_RESIDUAL_STREAM)):
        eng = _engine("bfloat16", setup["params"], 34, options=opt)
        assert eng.status(_lib.STATUS_RESIDUAL_SPLIT) == (1 if name == "split" else 0)
        eng.encode(x)
        ids, logits0 = eng.decode(num_steps=40, return_first_logits=True)
        out[name] = (ids.cpu().numpy(), logits0.cpu().numpy())
    a, b = out["split"][1], out["single"][1]
    # The two forms feed the SAME bf16 operands; only the row scale's summation order differs (32 partials vs an
    # in-kernel chain), i.e. by <= 1 ulp of f32.  That ulp can flip the bf16 rounding of one q/k/v element, which
    # moves that ROW's logits by ~1e-3 (measured on MI355X: 4 of 34 rows at 2e-3, the rest at <= 1.3e-7) -- a wrong
    # partial sum would instead shift EVERY row at the percent level.
    d = np.linalg.norm(a.astype(np.float64) - b, axis=1) / np.linalg.norm(b.astype(np.float64), axis=1)
    clean = d < 2e-6
    # (the 34 rows are copies of 3 segments, so flips come in groups: no fraction-of-rows bound; a wrong partial sum
    # would put EVERY row above 1e-2, a flip stays near 2e-3, identical operands give < 2e-7)
    assert clean.any(), d
    assert d.max() < 6e-3, d
    assert np.array_equal(out["split"][0][clean], out["single"][0][clean])


def test_folded_cross_query_projection_matches_the_separate_launch(setup):
    """bf16 decode folds the cross-attention q-projection into the QKV and self out-projection launches
    (y_new.Wq' = y_old.Wq' + attn.(Wo.Wq'), 1/rms applied by the cross-attention kernel from the partial sums): the
    same function of the same weights, rounded in different places.  Against the separate launch
    (options = MT3_OPT_SEPARATE_PROJECTIONS):
    step-0 logits within bf16 noise on every row; against the f32 oracle both stay inside the bf16 bound."""
    x = torch.from_numpy(np.repeat(setup["x"], 12, axis=0)[:34]).cuda()
    out = {}
    # (the q / k / v fold of round 3 has its own test, tests/test_gpu_parity_r3.py: here the cross-query fold alone)
    for name, opt in (("fold", _lib.OPT_SEPARATE_QKV_PROJECTION), ("separate", _lib.OPT_SEPARATE_PROJECTIONS)):
        eng = _engine("bfloat16", setup["params"], 34, options=opt)
        assert eng.status(_lib.STATUS_Q_FOLD) == (1 if name == "fold" else 0) and eng.status(_lib.STATUS_QKV_FOLD) == 0
        eng.encode(x)
        ids, logits0 = eng.decode(num_steps=24, return_first_logits=True)
        out[name] = (ids.cpu().numpy(), logits0.cpu().numpy())
    a, b = out["fold"][1], out["separate"][1]
    d = np.linalg.norm(a.astype(np.float64) - b, axis=1) / np.linalg.norm(b.astype(np.float64), axis=1)
    assert d.max() < 1e-2 and np.median(d) < 5e-3, d
    ref = np.repeat(setup["logits_ref"][:, 0], 12, axis=0)[:34]
    for got in (a, b):
        r = np.linalg.norm(got.astype(np.float64) - ref, axis=1) / np.linalg.norm(ref.astype(np.float64), axis=1)
        assert r.max() < 3e-2, r


def test_inference_model_end_to_end():
    """InferenceModel('random:0', 'mt3')(audio): product notes == oracle symbolic stage on the product's tokens."""
    from mt3_amd import inference
    from oracle import symbolic as S
    audio = OF.synth_audio(3, seed=5).reshape(-1)[: 2 * 32768 + 5000]       # 2 full segments + a short one
    m = inference.InferenceModel("random:0", "mt3", batch_size=4, early_exit=False)
    assert m.inputs_length == 256 and m.outputs_length == 1024 and m.batch_size == 4
    assert m.input_shapes == {"encoder_input_tokens": (4, 256), "decoder_input_tokens": (4, 1024)}
    ds = m.audio_to_dataset(audio)
    ex = m.preprocess(ds)
    assert len(ex) == 3 and ex[0]["inputs"].shape == (256, 512) and ex[2]["inputs"].shape[0] < 256
    # frontend rows of the examples match the oracle frontend on the same samples
    ref0 = OF.compute_logmel(audio[:32768], np.float64, tables="tf32")      # the product's default tables
    assert np.abs(ex[0]["inputs"] - ref0)[np.exp(ref0) > 1e-2].max() < 1e-3
    ns = m(audio)
    feats = np.zeros((3, 256, 512), np.float32)
    for i, e in enumerate(ex):
        feats[i, : e["inputs"].shape[0]] = e["inputs"]
    toks = m.predict_tokens({"encoder_input_tokens": feats})
    assert toks.shape == (3, 1024) and toks.dtype == np.int32
    preds = [m.postprocess(t, e) for t, e in zip(toks, ex)]
    oc = S.build_codec(S.VocabularyConfig(num_velocity_bins=1))
    ref = S.event_predictions_to_ns(preds, oc, "ties")["est_ns"]
    got = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument) for n in ns.notes]
    assert got == ref.as_tuples() and ns.total_time == ref.total_time
    with pytest.raises(ValueError):
        inference.InferenceModel("random:0", "nope")


def test_inference_edge_cases():
    """Boundaries of the host framing the reference tests only implicitly (NB:318-335): empty audio (one
    1-frame segment), an exact multiple of the segment length (the always-added hop spills into one more
    1-frame segment), more segments than `batch_size` (several engine batches == one big batch, row for row),
    and the C ABI's error path surfacing as an exception instead of a silent fallback."""
    from mt3_amd import _lib, inference
    small = dict(batch_size=2, early_exit=True, decoding="greedy", schedule="batch")     # the reference's loop (NB:295-301)
    m = inference.InferenceModel("random:0", "mt3", **small)
    ns = m(np.zeros(0, np.float32))
    ex = m.preprocess(m.audio_to_dataset(np.zeros(0, np.float32)))
    assert len(ex) == 1 and ex[0]["inputs"].shape == (1, 512) and ns.total_time >= 0.0
    audio = OF.synth_audio(2, seed=11).reshape(-1)                          # exactly 2 x 32768 samples
    ex = m.preprocess(m.audio_to_dataset(audio))
    assert [e["inputs"].shape[0] for e in ex] == [256, 256, 1]
    assert abs(ex[2]["input_times"][0] - 4.096) < 1e-12
    feats = np.zeros((3, 256, 512), np.float32)
    for i, e in enumerate(ex):
        feats[i, : e["inputs"].shape[0]] = e["inputs"]
    split = m.predict_tokens({"encoder_input_tokens": feats})               # batches of 2 + 1
    assert m.rows_per_engine_call == [2, 1]
    big = inference.InferenceModel("random:0", "mt3", batch_size=4, early_exit=True, decoding="greedy")
    whole = big.predict_tokens({"encoder_input_tokens": feats})             # one refilled engine call (the default)
    assert big.rows_per_engine_call == [3] and big.batch_size == 4
    assert np.array_equal(split, whole)
    # error path: decode before any encode, bad shapes
    eng = network.Transformer(network.T5Config(), input_length=256, max_decode_length=1024, max_batch=2)
    eng.load_params(network.init_random_params(network.T5Config(), seed=0))
    with pytest.raises(_lib.Mt3Error):
        eng._batch = 1
        eng.decode(num_steps=4)
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(1, 100, 512, device="cuda"))
    with pytest.raises(_lib.Mt3Error):
        eng.encode(torch.zeros(3, 256, 512, device="cuda"))               # batch > max_batch


def test_ismir2021_preset_and_base_shape():
    """The other reference presets: ismir2021 (T = 512 frames, 127 velocity bins, vocab 1664,
    NoteEncodingSpec) end to end, and the ismir2022/base.gin shape (emb 768, 12 heads, 12+12 layers,
    mlp 2048) encoder + first decode steps against the oracle (bf16 tolerances)."""
    from mt3_amd import inference
    from oracle import symbolic as S
    audio = OF.synth_audio(5, seed=9).reshape(-1)[: 2 * 65536 + 7000]        # 2 full 4.096 s segments + a short one
    m = inference.InferenceModel("random:1", "ismir2021", batch_size=2, early_exit=False)
    assert m.inputs_length == 512 and m.model_config.vocab_size == 1664
    assert m.codec.num_classes == 1514 and m.vocabulary.vocab_size == 1617
    ns = m(audio)
    ex = m.preprocess(m.audio_to_dataset(audio))
    assert [e["inputs"].shape[0] for e in ex[:2]] == [512, 512] and len(ex) == 3
    feats = np.zeros((3, 512, 512), np.float32)
    for i, e in enumerate(ex):
        feats[i, : e["inputs"].shape[0]] = e["inputs"]
    toks = m.predict_tokens({"encoder_input_tokens": feats})
    preds = [m.postprocess(t, e) for t, e in zip(toks, ex)]
    assert abs(preds[1]["start_time"] - 4.09) < 1e-9                       # 4.096 floored to the 10 ms grid
    ref = S.event_predictions_to_ns(preds, S.build_codec(S.VocabularyConfig(num_velocity_bins=127)), "notes")["est_ns"]
    got = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument) for n in ns.notes]
    assert got == ref.as_tuples()
    # engine vs oracle at T = 512 (bf16)
    params = network.init_random_params(m.model_config, seed=1)
    orc = _oracle(m.model_config, params)
    enc_ref = orc.encode(feats[:2]).numpy()
    enc = m.model.encode(torch.from_numpy(feats[:2]).cuda(), return_encoded=True).cpu().numpy()
    assert rel(enc, enc_ref) < 2e-2
    del m
    # ismir2022/base.gin shape
    base = network.T5Config(**{**{f: getattr(network.MT3_BASE, f) for f in network.MT3_BASE.__dataclass_fields__},
                               "dtype": "bfloat16"})
    pb = network.init_random_params(base, seed=2, norm_scale_jitter=0.1)
    assert sum(v.size for v in pb.values()) == 200_980_992
    eng = network.Transformer(base, input_length=T, max_decode_length=L, max_batch=2)
    eng.load_params(pb)
    x = _inputs(2, seed=4)
    enc = eng.encode(torch.from_numpy(x).cuda(), return_encoded=True).cpu().numpy()
    ids, logits0 = eng.decode(num_steps=4, return_first_logits=True)
    ob = _oracle(base, pb)
    enc_ref = ob.encode(x)
    _, lref = ob.greedy_decode(enc_ref, 1, return_logits=True)
    assert rel(enc, enc_ref.numpy()) < 3e-2, rel(enc, enc_ref.numpy())
    assert rel(logits0.cpu().numpy(), lref[:, 0].numpy()) < 4e-2


def test_full_size_properties():
    """BASELINE config 3 size (batch 256, bf16): size-independent properties instead of an oracle run.
    * segments are independent: decoding a permuted batch gives the permuted token rows, bit for bit;
    * the same segment repeated in every row decodes identically in every row (no cross-row leakage);
    * ids stay in [0, vocab) and `_decode_tf` + the host note decoder accept every row."""
    from mt3_amd import metrics_utils, note_sequences, spectrograms, synthetic, vocabularies
    B, steps = 256, 96
    cfg = network.T5Config(dtype="bfloat16")
    eng = network.Transformer(cfg, input_length=T, max_decode_length=L, max_batch=B)
    eng.load_params(network.init_random_params(cfg, seed=0))
    audio = synthetic.synth_audio(B, seed=11)
    logmel = spectrograms.compute_spectrogram_batch(audio, None)
    eng.encode(logmel)
    ids = eng.decode(num_steps=steps)
    perm = torch.randperm(B, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    eng.encode(logmel[perm])
    ids_p = eng.decode(num_steps=steps)
    assert torch.equal(ids_p, ids[perm]), "rows of a batch must not influence each other"
    eng.encode(logmel[:1].expand(B, -1, -1).contiguous())
    same = eng.decode(num_steps=steps)
    assert torch.equal(same, same[:1].expand(B, -1)) and torch.equal(same[0], ids[0])
    assert int(ids.min()) >= 0 and int(ids.max()) < cfg.vocab_size and bool((ids[:, steps:] == 0).all())
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    toks = vocabularies.vocabulary_from_codec(codec).decode_tf(ids).cpu().numpy()
    preds = [{"est_tokens": r[: np.argmax(r == -1)] if (r == -1).any() else r[:steps],
              "start_time": i * 2.048 - (i * 2.048) % 0.01} for i, r in enumerate(toks)]
    res = metrics_utils.event_predictions_to_ns(preds, codec, note_sequences.NoteEncodingWithTiesSpec)
    n_tok = sum(len(p["est_tokens"]) for p in preds)
    assert res["est_invalid_events"] + res["est_dropped_events"] <= n_tok
    # frontend at this size: every full segment row is finite, short segments get exact zero rows
    assert bool(torch.isfinite(logmel).all())
    n_frames = [256 if i % 3 else 100 for i in range(B)]
    lm2 = spectrograms.compute_spectrogram_batch(audio, n_frames)
    assert bool((lm2[1::3, 100:] != 0).any()) and bool((lm2[0::3, 100:] == 0).all())
    assert torch.equal(lm2[1], logmel[1]) and torch.equal(lm2[0, :100], spectrograms.compute_spectrogram_batch(
        audio[:1, : 100 * 128 + 28 * 128].contiguous(), [100])[0, :100])


def test_npz_checkpoint_round_trip(tmp_path):
    """restore_from_checkpoint from a flat .npz in the reference's parameter names == passing the dict."""
    from mt3_amd import inference
    cfg = network.T5Config(num_encoder_layers=2, num_decoder_layers=2)
    params = network.init_random_params(network.T5Config(**{**{f: getattr(cfg, f) for f in cfg.__dataclass_fields__}}),
                                        seed=4)
    path = tmp_path / "ckpt.npz"
    np.savez(path, **params)
    audio = OF.synth_audio(1, seed=2)[0][:20000]
    a = inference.InferenceModel(str(path), "mt3", config=cfg, batch_size=2, early_exit=False)
    b = inference.InferenceModel(params, "mt3", config=cfg, batch_size=2, early_exit=False)
    ta = a.predict_tokens({"encoder_input_tokens": np.stack([e["inputs"] if len(e["inputs"]) == 256 else np.pad(
        e["inputs"], ((0, 256 - len(e["inputs"])), (0, 0))) for e in a.preprocess(a.audio_to_dataset(audio))])})
    tb = b.predict_tokens({"encoder_input_tokens": np.stack([e["inputs"] if len(e["inputs"]) == 256 else np.pad(
        e["inputs"], ((0, 256 - len(e["inputs"])), (0, 0))) for e in b.preprocess(b.audio_to_dataset(audio))])})
    assert np.array_equal(ta, tb)
    # the t5x directory layout (msgpack index + zarr arrays) restores the same weights
    from mt3_amd import checkpoints
    checkpoints.save_t5x_checkpoint(str(tmp_path / "t5x_ckpt"), params, chunk_rows=100)
    c = inference.InferenceModel(str(tmp_path / "t5x_ckpt"), "mt3", config=cfg, batch_size=2, early_exit=False)
    tc = c.predict_tokens({"encoder_input_tokens": np.stack([e["inputs"] if len(e["inputs"]) == 256 else np.pad(
        e["inputs"], ((0, 256 - len(e["inputs"])), (0, 0))) for e in c.preprocess(c.audio_to_dataset(audio))])})
    assert np.array_equal(ta, tc)
    with pytest.raises(ValueError):
        inference.InferenceModel("/nonexistent/checkpoint_dir", "mt3", config=cfg)
    with pytest.raises(Exception):
        bad = dict(params)
        bad.pop("decoder/logits_dense/kernel")
        inference.InferenceModel(bad, "mt3", config=cfg)
