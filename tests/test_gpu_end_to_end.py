"""Audio -> notes, product against oracle, END TO END (round 5; VERDICT r4 "next" #3 and #6).

Both sides start from the same samples and share nothing after that: the product runs `InferenceModel.__call__`
(NB:283-308: frames -> segments -> log-mel kernel -> engine -> ids -> tokens -> C++ note decoding), the oracle its own
numpy frontend, torch-CPU f32 network, decode loop and pure-Python note state machine.  Random-init weights decode next
to no notes (58 in 262,144 tokens), so the logits columns of the tokens a note needs are boosted
(synthetic.boost_note_events): the streams below hold tens of notes.  f32 engine = the reference's precision.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import inference, network, synthetic  # noqa: E402
from oracle import frontend as OF, network as ON, symbolic as OS  # noqa: E402

CFG = dict(num_encoder_layers=2, num_decoder_layers=2)      # two layers each: the CPU side stays quick


# the two presets of the notebook class (NB:176-183): inputs_length, velocity bins, note-decoding mode of the oracle,
# codec classes, logits columns (vocabulary padded to a multiple of 128), boosts that make random weights decode notes
PRESETS = {
    "mt3": dict(T=256, bins=1, mode="ties", classes=1388, vocab=1536, boost=dict(eos=3.0)),
    # NoteEncodingSpec has no tie section (a `tie` token would only count as invalid): boost the 128 velocity tokens instead,
    # velocity 0 among them = the note-offs (mt3/note_sequences.py:341-362)
    "ismir2021": dict(T=512, bins=127, mode="notes", classes=1514, vocab=1664, boost=dict(eos=3.0, tie=1.0, velocity=2.0)),
}


def _oracle_notes(params, wav, decoding, T=256, hop=128, preset="mt3"):
    """the oracle's own audio -> notes: NB:318-335 framing, oracle/frontend.py, oracle/network.py, oracle/symbolic.py"""
    P = PRESETS[preset]
    w = np.pad(np.asarray(wav, np.float32), [0, hop - len(wav) % hop])
    frames = w.reshape(-1, hop)
    segs = [frames[i:i + T] for i in range(0, len(frames), T)]
    lm = np.zeros((len(segs), T, 512), np.float32)
    for i, sg in enumerate(segs):
        lm[i, : len(sg)] = OF.compute_logmel(sg.reshape(-1), np.float32, tables="tf32")[: len(sg)]
    orc = ON.Oracle(params, ON.T5Config(vocab_size=P["vocab"], **CFG))
    with torch.no_grad():
        enc = orc.encode(lm)
        if decoding == "greedy":
            ids, logits = orc.greedy_decode(enc, 1024, return_logits=True)
        else:
            ids, logits = orc.beam1_decode(enc, 1024), None
    toks = OS.GenericTokenVocabulary(P["classes"], extra_ids=100).decode_tf(ids)
    preds = [{"est_tokens": OS.trim_eos(t), "start_time": OS.floor_start_time(i * T * hop / 16000.0, 100)}
             for i, t in enumerate(toks)]
    res = OS.event_predictions_to_ns(preds, OS.build_codec(OS.VocabularyConfig(num_velocity_bins=P["bins"])), P["mode"])
    return res["est_ns"], toks, logits


def _tuples(ns):
    return [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, bool(n.is_drum), n.instrument) for n in ns.notes]


@pytest.mark.parametrize("preset,decoding", [("mt3", "beam1"), ("mt3", "greedy"), ("ismir2021", "beam1"),
                                             ("ismir2021", "greedy")])
def test_audio_to_notes_matches_the_oracles_own_audio_to_notes(preset, decoding):
    """Both presets of the notebook class (NB:176-183).  `ismir2021` = configs[0]'s preset: 512-frame (4.096 s) segments,
    127 velocity bins, NoteEncodingSpec without ties (mt3/note_sequences.py:416-446), 1664 logits columns."""
    P = PRESETS[preset]
    T = P["T"]
    cfg = network.T5Config(dtype="float32", vocab_size=P["vocab"], **CFG)
    params = synthetic.boost_note_events(network.init_random_params(cfg, seed=1, norm_scale_jitter=0.1),
                                         num_velocity_bins=P["bins"], **P["boost"])
    # 2 segments + a short one; generated on the CPU so that the samples -- and with them the notes the oracle decodes
    # from these weights -- are the same on every box
    wav = synthetic.synth_audio(3, seed=0, device="cpu", seg_samples=T * 128).reshape(-1)[: 2 * T * 128 + 9000].numpy()
    m = inference.InferenceModel(params, preset, config=cfg, decoding=decoding)
    assert m.inputs_length == T and m.model_config.vocab_size == P["vocab"] and m.codec.num_classes == P["classes"]
    ns = m(wav)
    ex = m.preprocess(m.audio_to_dataset(wav))
    batch = np.zeros((len(ex), T, 512), np.float32)
    for i, e in enumerate(ex):
        batch[i, : e["inputs"].shape[0]] = e["inputs"]
    toks = m.predict_tokens({"encoder_input_tokens": batch})
    ref_ns, ref_toks, logits = _oracle_notes(params, wav, decoding, T=T, preset=preset)
    for r in range(len(ex)):
        d = np.nonzero(toks[r] != ref_toks[r])[0]
        if d.size and logits is not None:          # a flip is only excusable on an oracle tie (< 2e-4 sigma)
            lg = logits[r, int(d[0])].double()
            top = torch.topk(lg, 2).values
            assert float((top[0] - top[1]) / lg.std()) < 2e-4, (r, int(d[0]), float((top[0] - top[1]) / lg.std()))
            pytest.skip("row %d diverges at step %d on an oracle tie" % (r, int(d[0])))
        assert not d.size, (preset, decoding, r, int(d[0]))
    assert len(ref_ns.notes) >= 5, "the boosted weights should decode notes"
    assert _tuples(ns) == ref_ns.as_tuples() and ns.total_time == ref_ns.total_time


def test_one_file_gives_the_same_notes_in_one_refilled_call_and_in_the_references_batches_of_8():
    """`InferenceModel.__call__` sizes its engine to the file (one mt3_engine_transcribe call, finished rows
    refilled) while `batch_size` stays the reference's 8; `schedule="batch"` is the reference's literal loop of 8-row
    batch-synchronous calls (NB:190,295-301).  Same notes; the log-mel never leaves the device in between."""
    cfg = network.T5Config(dtype="float32", **CFG)
    params = synthetic.boost_note_events(network.init_random_params(cfg, seed=1, norm_scale_jitter=0.1), eos=3.0)
    wav = synthetic.synth_audio(43, seed=4, device="cpu").reshape(-1)[: 42 * 32768 + 777].numpy()    # 43 segments: 88 s of audio
    a = inference.InferenceModel(params, "mt3", config=cfg, max_slots=32)
    b = inference.InferenceModel(params, "mt3", config=cfg, schedule="batch")
    na, nb = a(wav), b(wav)
    assert a.batch_size == 8 and a.input_shapes["encoder_input_tokens"] == (8, 256)
    assert a.engine_slots == 32 and a.rows_per_engine_call == [43] and a.model.transcribe_stats["refills"] == 11
    assert b.engine_slots == 8 and b.rows_per_engine_call == [8, 8, 8, 8, 8, 3]
    assert len(na.notes) >= 20
    assert _tuples(na) == _tuples(nb) and na.total_time == nb.total_time


def test_several_files_as_one_job_give_each_file_its_own_notes():
    """`transcribe_many`: the segments of five files of different lengths share the engine's slots in one refilled call;
    every file's notes equal what `model(audio)` returns for it alone."""
    cfg = network.T5Config(dtype="float32", **CFG)
    params = synthetic.boost_note_events(network.init_random_params(cfg, seed=1, norm_scale_jitter=0.1), eos=3.0)
    audio = synthetic.synth_audio(30, seed=6, device="cpu").reshape(-1).numpy()
    cuts = [0, 5 * 32768 + 100, 5 * 32768 + 100 + 40000, 17 * 32768, 17 * 32768 + 128, 29 * 32768 + 5555]
    files = [audio[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    m = inference.InferenceModel(params, "mt3", config=cfg, max_slots=16)
    together = m.transcribe_many(files)
    assert m.rows_per_engine_call == [sum(-(-(len(f) // 128 + 1) // 256) for f in files)] and m.engine_slots == 16
    assert len(together) == len(files) and sum(len(ns.notes) for ns in together) >= 20
    for f, ns in zip(files, together):
        alone = m(f)
        assert _tuples(ns) == _tuples(alone) and ns.total_time == alone.total_time
    assert m.transcribe_many([]) == []
