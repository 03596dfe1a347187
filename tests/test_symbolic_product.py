"""The product's host symbolic stage (libmt3hip.so: csrc/symbolic.cpp through the C ABI and the
Python mirror) against (a) the reference's unit-test literals, (b) golden vectors produced by the
reference's real modules, (c) the oracle on random token soup.  Integer fields bit-exact, times
exact (same double expressions).  CPU-only: this stage has no device code."""
import json
import os

import numpy as np
import pytest

from mt3_amd import event_codec as EC, metrics_utils as MU, note_sequences as NS, vocabularies as V
from oracle import symbolic as S
from tests import symbolic_cases as K

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "symbolic_golden.json")))
SPEC = {"onsets": NS.NoteOnsetEncodingSpec, "notes": NS.NoteEncodingSpec, "ties": NS.NoteEncodingWithTiesSpec}


def _codec(ranges, max_shift=100, sps=100):
    return EC.Codec(max_shift, sps, [EC.EventRange(*r) for r in ranges])


def _tuples(ns):
    return [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ns.notes]


def test_event_codec_literals():
    ec = _codec([("pitch", 0, 127)])
    ev = [EC.Event("pitch", 60), EC.Event("shift", 5), EC.Event("pitch", 62)]
    enc = [ec.encode_event(e) for e in ev]
    assert enc == [161, 5, 163]
    assert [ec.decode_event_index(i) for i in enc] == ev
    assert ec.max_shift_steps == 100
    assert not ec.is_shift_event_index(-1) and ec.is_shift_event_index(100) and not ec.is_shift_event_index(101)
    with pytest.raises(ValueError):
        ec.decode_event_index(229)
    with pytest.raises(ValueError):
        ec.encode_event(EC.Event("pitch", 128))
    with pytest.raises(ValueError):
        ec.encode_event(EC.Event("drum", 1))


def test_vocabulary_host_literals():
    assert V.velocity_to_bin(0, 1) == 0 and V.bin_to_velocity(0, 127) == 0
    for b in range(1, 128):
        assert V.velocity_to_bin(V.bin_to_velocity(b, 127), 127) == b
    v = V.GenericTokenVocabulary(32)
    assert v.encode([1, 2, 3]) == [4, 5, 6]
    with pytest.raises(ValueError):
        v.encode([-1, 15, 31])
    with pytest.raises(ValueError):
        v.encode([0, 15, 32])


@pytest.mark.parametrize("case", K.SINGLE, ids=lambda c: c["name"])
def test_decode_events_literals(case):
    codec = _codec(case["ranges"])
    ns, inv, drop = MU.decode_events_single(case["tokens"], case["start"], case["max_time"], codec, SPEC[case["mode"]])
    assert (inv, drop) == (case["invalid"], case["dropped"])
    got = _tuples(ns)
    assert len(got) == len(case["notes"])
    for g, e in zip(got, case["notes"]):
        assert g[2:] == e[2:]
        assert abs(g[0] - e[0]) < 1e-12 and abs(g[1] - e[1]) < 1e-12
    assert abs(ns.total_time - case["total"]) < 1e-12
    if "instruments" in case:
        assert [n.instrument for n in ns.notes] == case["instruments"]


@pytest.mark.parametrize("case", K.COMBINE, ids=lambda c: c["name"])
def test_combiner_literals(case):
    codec = _codec(case["ranges"])
    preds = [{"start_time": st, "est_tokens": toks, "raw_inputs": [i, i]} for i, (st, toks) in enumerate(case["segments"])]
    res = MU.event_predictions_to_ns(preds, codec, SPEC[case["mode"]])
    assert res["est_invalid_events"] == case["invalid"] and res["est_dropped_events"] == case["dropped"]
    got = _tuples(res["est_ns"])
    assert len(got) == len(case["notes"])
    for g, e in zip(got, case["notes"]):
        assert g[2:] == e[2:]
        assert abs(g[0] - e[0]) < 1e-12 and abs(g[1] - e[1]) < 1e-12
    assert abs(res["est_ns"].total_time - case["total"]) < 1e-12
    np.testing.assert_array_equal(res["raw_inputs"], [0, 0, 1, 1, 2, 2])   # metrics_utils_test.py:82
    if "instruments" in case:
        assert [n.instrument for n in res["est_ns"].notes] == case["instruments"]


@pytest.mark.parametrize("preset", ["mt3", "ismir2021"])
def test_codec_tables_vs_reference(preset):
    g = GOLD["codecs"][preset]
    codec = V.build_codec(V.VocabularyConfig(num_velocity_bins=g["num_velocity_bins"]))
    vocab = V.vocabulary_from_codec(codec)
    assert codec.num_classes == g["num_classes"]
    assert vocab.vocab_size == g["vocab_size"] and V.num_embeddings(vocab) == g["num_embeddings"]
    for t, (lo, hi) in g["type_ranges"].items():
        assert codec.event_type_range(t) == (lo, hi)
    for idx, t, val in g["decode_probe"]:
        ev = codec.decode_event_index(idx)
        assert (ev.type, ev.value) == (t, val) and codec.encode_event(ev) == idx


@pytest.mark.parametrize("i", range(len(GOLD["decode_cases"])))
def test_decode_cases_vs_reference(i):
    c = GOLD["decode_cases"][i]
    codec = V.build_codec(V.VocabularyConfig(num_velocity_bins=c["num_velocity_bins"]))
    preds = [{"start_time": s["start_time"], "est_tokens": np.array(s["tokens"], np.int32)} for s in c["segments"]]
    res = MU.event_predictions_to_ns(preds, codec, SPEC[c["mode"]])
    assert res["est_invalid_events"] == c["invalid"] and res["est_dropped_events"] == c["dropped"]
    got = [[n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument]
           for n in res["est_ns"].notes]
    assert got == c["notes"]
    assert res["est_ns"].total_time == c["total_time"]


@pytest.mark.parametrize("seed", range(8))
def test_product_vs_oracle_random(seed):
    """Larger random streams (1024-token rows, the path's maximum) -- product == oracle exactly."""
    rng = np.random.default_rng(seed)
    bins = 1 if seed % 2 == 0 else 127
    mode = ("ties", "notes", "onsets")[seed % 3]
    pc = V.build_codec(V.VocabularyConfig(num_velocity_bins=bins))
    oc = S.build_codec(S.VocabularyConfig(num_velocity_bins=bins))
    preds = []
    for s in range(int(rng.integers(1, 12))):
        n = int(rng.choice([0, 1, 17, 300, 1024]))
        kinds = rng.random(n)
        toks = np.where(kinds < 0.3, rng.integers(1, 40, n), rng.integers(-3, pc.num_classes + 10, n)).astype(np.int32)
        preds.append({"est_tokens": toks, "start_time": S.floor_start_time(s * 2.048, 100)})
    order = rng.permutation(len(preds))
    preds = [preds[i] for i in order]
    a = MU.event_predictions_to_ns(preds, pc, SPEC[mode])
    b = S.event_predictions_to_ns(preds, oc, mode)
    assert a["est_invalid_events"] == b["est_invalid_events"] and a["est_dropped_events"] == b["est_dropped_events"]
    ga = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum, n.instrument) for n in a["est_ns"].notes]
    assert ga == b["est_ns"].as_tuples()
    assert a["est_ns"].total_time == b["est_ns"].total_time and a["start_times"] == b["start_times"]


def test_empty_inputs():
    codec = V.build_codec(V.VocabularyConfig(num_velocity_bins=1))
    res = MU.event_predictions_to_ns([], codec, NS.NoteEncodingWithTiesSpec)
    assert res["est_ns"].notes == [] and res["est_invalid_events"] == 0
    res = MU.event_predictions_to_ns([{"est_tokens": np.zeros(0, np.int32), "start_time": 0.0}], codec,
                                     NS.NoteEncodingWithTiesSpec)
    assert res["est_ns"].notes == [] and res["est_ns"].total_time == 0.0
