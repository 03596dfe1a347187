"""Pins oracle/symbolic.py: reference unit-test literals + golden vectors made by
the reference's real modules (tests/golden/make_symbolic_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import symbolic as S
from tests import symbolic_cases as K

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "symbolic_golden.json")))


def _codec(ranges, max_shift=100, sps=100):
    return S.Codec(max_shift, sps, ranges)


def test_event_codec_literals():            # event_codec_test.py:26-52
    ec = _codec([("pitch", 0, 127)])
    ev = [S.Event("pitch", 60), S.Event("shift", 5), S.Event("pitch", 62)]
    enc = [ec.encode_event(e) for e in ev]
    assert enc == [161, 5, 163]
    assert [ec.decode_event_index(i) for i in enc] == ev
    assert ec.max_shift_steps == 100
    assert not ec.is_shift_event_index(-1) and ec.is_shift_event_index(0)
    assert ec.is_shift_event_index(100) and not ec.is_shift_event_index(101)


def test_vocabulary_literals():             # vocabularies_test.py:28-102
    assert S.velocity_to_bin(0, 1) == 0 and S.velocity_to_bin(0, 127) == 0
    assert S.bin_to_velocity(0, 1) == 0 and S.bin_to_velocity(0, 127) == 0
    assert S.velocity_to_bin(S.bin_to_velocity(1, 1), 1) == 1
    for b in range(1, 128):
        assert S.velocity_to_bin(S.bin_to_velocity(b, 127), 127) == b
    v = S.GenericTokenVocabulary(32)
    assert v.encode([1, 2, 3]) == [4, 5, 6]
    assert v.decode([4, 5, 6]) == [1, 2, 3]
    np.testing.assert_array_equal(v.decode_tf(np.array([4, 5, 6], np.int32)), [1, 2, 3])
    v4 = S.GenericTokenVocabulary(32, extra_ids=4)
    assert v4.decode([0, 2, 3, 4, 34, 35]) == [-2, -2, 0, 1, 31, -2]
    np.testing.assert_array_equal(v4.decode_tf(np.array([0, 2, 3, 4, 34, 35])), [-2, -2, 0, 1, 31, -2])
    enc = [0, 2, 3, 4, 1, 0, 1, 0]
    assert v.decode(enc) == [-2, -2, 0, 1, -1]
    np.testing.assert_array_equal(v.decode_tf(np.array(enc)), [-2, -2, 0, 1, -1, -1, -1, -1])
    v.encode([0, 15, 31])
    with pytest.raises(ValueError):
        v.encode([-1, 15, 31])
    with pytest.raises(ValueError):
        v.encode([0, 15, 32])
    assert v.decode_tf(np.array([3, 4], np.int64)).dtype == np.int64


@pytest.mark.parametrize("case", K.SINGLE, ids=lambda c: c["name"])
def test_decode_events_literals(case):      # note_sequences_test.py:290-501
    codec = _codec(case["ranges"])
    dec = S.NoteDecoder(case["mode"])
    inv, drop = S.decode_events(dec, case["tokens"], case["start"], case["max_time"], codec)
    ns = dec.flush()
    if case["mode"] == "onsets":
        S.assign_instruments(ns)
    assert (inv, drop) == (case["invalid"], case["dropped"])
    got = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ns.notes]
    assert len(got) == len(case["notes"])
    for g, e in zip(got, case["notes"]):
        assert g[2:] == e[2:]
        assert abs(g[0] - e[0]) < 1e-12 and abs(g[1] - e[1]) < 1e-12
    assert abs(ns.total_time - case["total"]) < 1e-12
    if "instruments" in case:
        assert [n.instrument for n in ns.notes] == case["instruments"]


@pytest.mark.parametrize("case", K.COMBINE, ids=lambda c: c["name"])
def test_combiner_literals(case):           # metrics_utils_test.py:28-238
    codec = _codec(case["ranges"])
    preds = [{"start_time": st, "est_tokens": toks} for st, toks in case["segments"]]
    res = S.event_predictions_to_ns(preds, codec, case["mode"])
    ns = res["est_ns"]
    assert res["est_invalid_events"] == case["invalid"]
    assert res["est_dropped_events"] == case["dropped"]
    got = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ns.notes]
    assert len(got) == len(case["notes"])
    for g, e in zip(got, case["notes"]):
        assert g[2:] == e[2:]
        assert abs(g[0] - e[0]) < 1e-12 and abs(g[1] - e[1]) < 1e-12
    assert abs(ns.total_time - case["total"]) < 1e-12
    if "instruments" in case:
        assert [n.instrument for n in ns.notes] == case["instruments"]


@pytest.mark.parametrize("preset", ["mt3", "ismir2021"])
def test_codec_tables_vs_reference(preset):
    g = GOLD["codecs"][preset]
    codec = S.build_codec(S.VocabularyConfig(num_velocity_bins=g["num_velocity_bins"]))
    vocab = S.vocabulary_from_codec(codec)
    assert codec.num_classes == g["num_classes"]
    assert vocab.vocab_size == g["vocab_size"]
    assert S.num_embeddings(vocab) == g["num_embeddings"]
    for t, (lo, hi) in g["type_ranges"].items():
        assert codec.event_type_range(t) == (lo, hi)
    for idx, t, val in g["decode_probe"]:
        ev = codec.decode_event_index(idx)
        assert (ev.type, ev.value) == (t, val)
        assert codec.encode_event(ev) == idx
    # reference `_decode` is the per-id map (no EOS truncation): compare elementwise
    assert [vocab._map_one(i) for i in g["vocab_decode_in"]] == g["vocab_decode_out"]
    for vel, b, back in g["velocity_roundtrip"]:
        assert S.velocity_to_bin(vel, g["num_velocity_bins"]) == b
        assert S.bin_to_velocity(b, g["num_velocity_bins"]) == back


@pytest.mark.parametrize("i", range(len(GOLD["decode_cases"])))
def test_decode_cases_vs_reference(i):
    """Bit-exact against metrics_utils.event_predictions_to_ns run on the reference."""
    c = GOLD["decode_cases"][i]
    codec = S.build_codec(S.VocabularyConfig(num_velocity_bins=c["num_velocity_bins"]))
    preds = [{"start_time": s["start_time"], "est_tokens": np.array(s["tokens"], np.int32)}
             for s in c["segments"]]
    res = S.event_predictions_to_ns(preds, codec, c["mode"])
    assert res["est_invalid_events"] == c["invalid"]
    assert res["est_dropped_events"] == c["dropped"]
    got = [list(t) for t in res["est_ns"].as_tuples()]
    assert got == c["notes"]                 # float64 start/end compared exactly
    assert res["est_ns"].total_time == c["total_time"]


def test_trim_and_floor():
    np.testing.assert_array_equal(S.trim_eos([5, 6, -1, 7, -1]), [5, 6])
    np.testing.assert_array_equal(S.trim_eos([5, 6]), [5, 6])
    assert S.floor_start_time(2.048, 100) == 2.048 - 2.048 % 0.01
    assert abs(S.floor_start_time(2.048, 100) - 2.04) < 1e-9


def test_audio_framing():
    fr, t = S.audio_to_frames(np.ones(256, np.float32))
    assert fr.shape == (3, 128) and fr[2].sum() == 0 and t[1] == 1 / 125.0
    fr, t = S.audio_to_frames(np.ones(300, np.float32))
    assert fr.shape == (3, 128) and fr[2].sum() == 300 - 256
    segs = S.split_segments(np.zeros((600, 128)), np.arange(600) / 125.0, 256)
    assert [len(a) for a, _ in segs] == [256, 256, 88]
