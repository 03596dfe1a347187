"""Host-side pieces of the round-6 evaluation fixtures (no GPU): the synthetic music generator / renderer, the training
targets of tools/train_synthetic.py (the reference's preprocessor chain, mt3/tasks.py:142-178, through the encode side of
the codec) as an encode -> decode ROUND TRIP, the stub token rows of the multi-rank tests, the compact checkpoint format,
and the note-level comparison itself."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from mt3_amd import checkpoints, evaluation, metrics_utils, network, note_sequences, synthetic, vocabularies  # noqa: E402


def test_random_music_is_deterministic_and_well_formed():
    a, b = synthetic.random_music(20.0, seed=3), synthetic.random_music(20.0, seed=3)
    assert a == b and len(a.notes) > 60 and synthetic.random_music(20.0, seed=4) != a
    assert all(0.0 <= n.start_time < n.end_time <= 20.0 and n.end_time - n.start_time >= 0.1 for n in a.notes)
    assert all(36 <= n.pitch <= 96 and n.program == 0 and not n.is_drum for n in a.notes)
    # no two sounding notes of the same pitch, never more than six at once
    ev = sorted([(n.start_time, 1, n.pitch) for n in a.notes] + [(n.end_time, -1, n.pitch) for n in a.notes])
    sounding, most = {}, 0
    for _, d, p in ev:
        sounding[p] = sounding.get(p, 0) + d
        assert sounding[p] in (0, 1)
        most = max(most, sum(sounding.values()))
    assert most <= 6


def test_render_notes_puts_the_energy_where_the_notes_are():
    torch = pytest.importorskip("torch")
    sr = 16000
    wav = synthetic.render_notes([0.25, 0.5], [0.75, 0.9], [69, 57], [1.0, 0.6], [0, 1], 2, sr, seed=1, device="cpu").numpy()
    assert wav.shape == (2, sr) and wav.dtype == np.float32 and abs(np.abs(wav).max(1) - 0.9).max() < 1e-6
    for row, f0, on, off in ((0, 440.0, 0.25, 0.75), (1, 220.0, 0.5, 0.9)):
        seg = wav[row, int((on + 0.05) * sr): int((on + 0.05) * sr) + 4096]
        spec = np.abs(np.fft.rfft(seg * np.hanning(4096)))
        assert abs(np.argmax(spec) * sr / 4096 - f0) < 8.0                                   # the fundamental dominates
        assert np.abs(wav[row, : int(on * sr) - 16]).max() < 0.05 and np.abs(wav[row, int((off + 0.06) * sr):]).max() < 0.05
    # the same notes rendered as one file or as two: scatter-add by file index
    one = synthetic.render_notes([0.25], [0.75], [69], [1.0], [0], 1, sr, seed=1, device="cpu", noise=0.0).numpy()
    assert one.shape == (1, sr) and np.abs(one[0, int(0.3 * sr): int(0.7 * sr)]).max() > 0.3


def test_training_targets_round_trip_through_the_products_note_decoder():
    """truth notes -> per-segment target ids (train_synthetic.file_targets) -> decode_tf -> mt3_notes_decode == the truth
    quantised to the 10 ms grid: the pairs the fixture checkpoint was trained on say what the audio contains"""
    import train_synthetic as TS
    codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=1))
    vocab = vocabularies.vocabulary_from_codec(codec)
    ns = synthetic.random_music(4 * 2.048, seed=11, notes_per_second=7.0)
    tgt = TS.file_targets(ns, codec, 4 * 256)
    assert len(tgt) == 4 and all(t[-1] == 1 and (t[:-1] >= 3).all() for t in tgt)
    assert tgt[0][0] - 3 == codec.encode_event(vocabularies.event_codec.Event("tie", 0))     # segment 0: empty tie section
    assert any(t[0] - 3 != codec.encode_event(vocabularies.event_codec.Event("tie", 0)) for t in tgt[1:]), "ties expected"
    rows = np.full((4, 256), 0, np.int32)
    for i, t in enumerate(tgt):
        rows[i, : len(t)] = t
    toks = vocab.decode_tf(rows)
    starts = [s * 2.048 - (s * 2.048) % 0.01 for s in range(4)]
    rec, inv, drop, total = metrics_utils.decode_token_rows(codec, note_sequences.NoteEncodingWithTiesSpec, toks, starts)
    assert inv == 0 and drop == 0 and len(rec) == len(ns.notes)
    got = sorted((int(r["pitch"]), float(r["start_time"]), float(r["end_time"])) for r in rec)
    want = sorted((n.pitch, n.start_time, n.end_time) for n in ns.notes)
    for (p, a, b), (q, c, d) in zip(got, want):
        assert p == q and abs(a - c) <= 0.0101 and abs(b - d) <= 0.0101, ((p, a, b), (q, c, d))
    # ... and scored as a transcription it is perfect
    est = metrics_utils.note_sequence_from_records(rec, total)
    sc = evaluation.note_divergence(ns, est)
    assert sc["onset_f1_note_number"] == 1.0 and sc["onset_offset_f1_note_number"] == 1.0


def test_stub_token_rows_depend_on_the_segment_index_alone():
    a = synthetic.stub_token_rows(0, 600, 1024)
    b = np.concatenate([synthetic.stub_token_rows(0, 250, 1024), synthetic.stub_token_rows(250, 350, 1024)])
    assert a.dtype == np.int32 and np.array_equal(a, b)
    n = np.where((a == -1).any(1), (a == -1).argmax(1), 1024)
    assert 200 < n.mean() < 400 and (a[np.arange(600), n - 1] >= 0).all()
    assert not np.array_equal(a[0], a[256]) and np.array_equal(a[0] == -1, a[256] == -1)     # next file: transposed


def test_compact_checkpoint_round_trip_and_error(tmp_path):
    cfg = network.T5Config(emb_dim=64, num_heads=2, head_dim=16, mlp_dim=96, num_encoder_layers=1, num_decoder_layers=1,
                           vocab_size=40, input_depth=24)
    params = network.init_random_params(cfg, seed=2, norm_scale_jitter=0.1)
    path = str(tmp_path / "c.npz")
    deq = checkpoints.save_compact_npz(path, params, {"steps": 12})
    got = checkpoints.load_compact_npz(path)
    assert set(got) == set(params) and checkpoints.compact_npz_meta(path) == {"steps": 12}
    for k, w in params.items():
        assert got[k].dtype == np.float32 and np.array_equal(got[k], deq[k])
        if w.ndim == 2:                                        # int8 with a scale per output column (embedding: per row)
            axis = 1 if k.endswith("/embedding") else 0
            assert (np.abs(got[k] - w) <= np.abs(w).max(axis=axis, keepdims=True) / 254.0 + 1e-7).all(), k
        else:
            assert np.array_equal(got[k], w)
    assert os.path.getsize(path) < 0.4 * sum(w.size * 4 for w in params.values())
    np.savez(str(tmp_path / "plain.npz"), **params)
    with pytest.raises(checkpoints.CheckpointError):
        checkpoints.load_compact_npz(str(tmp_path / "plain.npz"))


def test_the_repository_checkpoint_is_the_mt3_tree():
    """tests/golden/mt3_synthetic_ckpt.npz: every parameter of mt3/gin/model.gin's network, nothing else"""
    path = os.path.join(ROOT, "tests", "golden", "mt3_synthetic_ckpt.npz")
    got = checkpoints.load_compact_npz(path)
    want = network.param_shapes(network.T5Config())
    assert set(got) == set(want) and all(got[k].shape == tuple(want[k]) for k in want)
    meta = checkpoints.compact_npz_meta(path)
    assert meta["steps"] >= 5000 and meta["held_out_token_accuracy_int8_weights"] > 0.85
    assert abs(meta["held_out_token_accuracy_int8_weights"] - meta["held_out_token_accuracy_f32_weights"]) < 0.005


def test_note_divergence_report():
    ns = synthetic.random_music(10.0, seed=5)
    same = evaluation.note_divergence(ns, ns)
    assert same["notes_identical"] and same["onset_f1_note_number"] == 1.0 and same["onset_offset_f1_hz"] == 1.0
    shifted = note_sequences.NoteSequence(notes=[note_sequences.Note(n.start_time + 0.2, n.end_time + 0.2, n.pitch, n.velocity)
                                                 for n in ns.notes])
    far = evaluation.note_divergence(ns, shifted)
    assert not far["notes_identical"] and far["onset_f1_note_number"] < 0.5
    # pitch 0 is a valid MIDI note but not a frequency mir_eval accepts: set aside and counted, not an exception
    z = note_sequences.NoteSequence(notes=list(ns.notes) + [note_sequences.Note(1.0, 2.0, 0, 100)])
    rep = evaluation.note_divergence(z, z)
    assert rep["pitch_0_notes_set_aside"] == 2 and rep["onset_f1_note_number"] == 1.0
    tok = np.array([[5, 6, -1, -1], [7, -1, -1, -1]], np.int32)
    rep = evaluation.note_divergence(ns, ns, tok, tok)
    assert rep["identical_rows_frac"] == 1.0 and rep["mean_tokens_per_row"] == 1.5
