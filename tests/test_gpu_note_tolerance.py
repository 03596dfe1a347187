"""Note-level tolerance of every engine precision bench.py times (BASELINE north_star: "decoded note
onsets/offsets within a stated fp tolerance"; SURVEY.md 8(d): onset +-50 ms, offset max(50 ms, 20 %), the mir_eval rule of
mt3/metrics.py:255-290, bars F1 >= 0.99 for bf16 and >= 0.97 for fp8).

Two sets of weights, one 5-minute file each through the drop-in class (`InferenceModel`, NB:283-308), f32 engine = reference:

* TRAINED: tests/golden/mt3_synthetic_ckpt.npz -- the MT3 network (mt3/gin/model.gin shape) trained for 14 minutes on one
  MI355X by tools/train_synthetic.py on synthetic music (int8 + per-column scales in the repository; the checkpoint IS
  the de-quantised f32 weights).  Its distributions are peaked and conditioned on the audio, which is what the tolerance
  question is about; the piece's ground-truth notes are known, so the same run also yields an ACCURACY figure.
  STATED BOUNDS (measured on the 10-minute bench file in brackets): bf16 onset F1 >= 0.99 [0.9995], onset + offset >= 0.99
  [0.998]; e4m3 K/V caches >= 0.97 [0.998 / 0.991]; + MXFP8 encoder >= 0.97 [0.995 / 0.979].
* BOOSTED random-init weights (synthetic.boost_note_events): tens of thousands of notes, but flat distributions -- one
  flipped arg-max re-rolls the rest of a row.  These MISS SURVEY's bars and are published as measured, with floors a
  margin below the measurement so that a regression still shows: bf16 0.85, e4m3 caches 0.72, + MXFP8 0.54 (10-minute file).

Also here: the trained weights through the ORACLE (frontend -> network -> beam-1 -> note state machine on the CPU) against
the f32 engine, and the same weights restored from a t5x-layout DIRECTORY against the .npz path.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from mt3_amd import checkpoints, evaluation, inference, network, synthetic  # noqa: E402

CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mt3_synthetic_ckpt.npz")
F32 = network.T5Config(dtype="float32")


@pytest.fixture(scope="module")
def trained():
    return checkpoints.load_compact_npz(CKPT)


def test_trained_checkpoint_reduced_precision_engines_within_the_stated_note_tolerance(trained):
    truth, wav = synthetic.synth_music(300.0, seed=78)                       # 5 minutes, ~1,550 notes; not the bench's piece
    rep = evaluation.compare_engines(trained, wav, F32, truth=truth)
    print({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != "vs_truth"}
           for k, v in rep.items()})
    assert rep["f32"]["notes"] > 1200
    acc = rep["f32"]["vs_truth"]
    assert acc["onset_f1_note_number"] >= 0.95 and acc["onset_offset_f1_note_number"] >= 0.90, acc   # it transcribes
    for mode, onset, both in (("bf16", 0.99, 0.99), ("fp8_kv", 0.97, 0.97), ("fp8_kv_mx8", 0.97, 0.97)):
        r = rep[mode]
        assert "error" not in r, r
        assert r["onset_f1_note_number"] >= onset and r["onset_f1_hz"] >= onset, (mode, r)
        assert r["onset_offset_f1_note_number"] >= both, (mode, r)
        # and no precision costs accuracy against the ground truth beyond noise
        assert r["vs_truth"]["onset_f1_note_number"] >= acc["onset_f1_note_number"] - 0.01, (mode, r["vs_truth"], acc)


def test_boosted_random_weights_are_published_as_measured():
    """flat distributions: the SURVEY bars are MISSED (stated in the module docstring and INTEGRATION.md); floors only"""
    n_seg = 147
    wav = synthetic.synth_audio(n_seg, seed=77, tones=6).reshape(-1)[: int(300.0 * 16000)].cpu().numpy()
    params = synthetic.boost_note_events(network.init_random_params(F32, seed=0), eos=4.0)
    rep = evaluation.compare_engines(params, wav, F32)
    print({k: {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in rep.items()})
    assert rep["f32"]["notes"] > 10_000
    for mode, floor in (("bf16", 0.70), ("fp8_kv", 0.55), ("fp8_kv_mx8", 0.35)):
        assert rep[mode]["onset_f1_note_number"] >= floor, (mode, rep[mode])
        assert rep[mode]["onset_f1_note_number"] < 0.99, "flat distributions should not pass for peaked ones"


def test_trained_checkpoint_f32_engine_matches_the_oracles_own_audio_to_notes(trained):
    """the f32 engine IS the reference for the tolerance figures above: hold it against the oracle on these weights too"""
    from oracle import frontend as OF, network as ON, symbolic as OS
    truth, wav = synthetic.synth_music(12 * 2.048 + 0.7, seed=5, device="cpu")              # 13 segments, the last one short
    m = inference.InferenceModel(trained, "mt3", dtype="float32")
    ns, toks = evaluation.file_notes_and_tokens(m, wav)
    w = np.pad(wav, [0, 128 - len(wav) % 128]).reshape(-1, 128)
    segs = [w[i:i + 256] for i in range(0, len(w), 256)]
    lm = np.zeros((len(segs), 256, 512), np.float32)
    for i, sg in enumerate(segs):
        lm[i, : len(sg)] = OF.compute_logmel(sg.reshape(-1), np.float32, tables="tf32")[: len(sg)]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    orc = ON.Oracle(trained, ON.T5Config())
    with torch.no_grad():
        ids_ref = orc.beam1_decode(orc.encode(lm), 1024)                   # (a row stops when t5x's bound says so)
    ref_toks = OS.GenericTokenVocabulary(1388, extra_ids=100).decode_tf(ids_ref)
    n_tok = [int(np.argmax(t == -1)) for t in ref_toks]
    assert all((t == -1).any() for t in ref_toks) and max(n_tok) < 200, "every row should end of its own accord"
    assert np.array_equal(toks, ref_toks), "tokens: f32 engine == oracle on the trained weights"
    preds = [{"est_tokens": OS.trim_eos(t), "start_time": OS.floor_start_time(i * 2.048, 100)} for i, t in enumerate(ref_toks)]
    ref_ns = OS.event_predictions_to_ns(preds, OS.build_codec(OS.VocabularyConfig(num_velocity_bins=1)), "ties")["est_ns"]
    got = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument) for n in ns.notes]
    assert len(got) >= 60 and got == ref_ns.as_tuples()
    sc = evaluation.note_divergence(truth, ns)
    assert sc["onset_f1_note_number"] >= 0.9, sc


def test_t5x_layout_directory_restores_the_same_engine(trained, tmp_path):
    """N1: the checkpoint written as a t5x directory (msgpack index + one zarr array per parameter, chunked along axis 0 as
    t5x shards them, small leaves inline) -> `InferenceModel(directory)` -> same tokens as the dict / .npz paths"""
    d = str(tmp_path / "checkpoint_7866")
    checkpoints.save_t5x_checkpoint(d, trained, step=7866, inline_below=1024, chunk_rows=256)
    _, wav = synthetic.synth_music(8 * 2.048, seed=11, device="cpu")
    a = inference.InferenceModel(d, "mt3")                                    # directory (NB:247-261)
    b = inference.InferenceModel(CKPT, "mt3")                                 # compact .npz
    c = inference.InferenceModel(trained, "mt3")                              # dict
    na, ta = evaluation.file_notes_and_tokens(a, wav)
    nb, tb = evaluation.file_notes_and_tokens(b, wav)
    nc, tc = evaluation.file_notes_and_tokens(c, wav)
    assert np.array_equal(ta, tb) and np.array_equal(ta, tc) and len(na.notes) >= 30
    assert [(n.start_time, n.end_time, n.pitch) for n in na.notes] == [(n.start_time, n.end_time, n.pitch) for n in nb.notes]
    assert a.model.ignored_params == []
