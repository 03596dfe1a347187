"""Encode side of the symbolic stage in the product mirror: bit-exact against the
reference's unit-test literals and against goldens produced by the reference's real
`encode_and_index_events` (tests/golden/make_symbolic_golden.py); plus the round trip
notes -> tokens -> notes through the product decoder."""
import json
import os

import numpy as np
import pytest

from mt3_amd import event_codec as EC, metrics_utils as MU, note_sequences as NS, run_length_encoding as RLE
from mt3_amd import vocabularies as V

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "symbolic_golden.json")))
SPEC = {"onsets": NS.NoteOnsetEncodingSpec, "notes": NS.NoteEncodingSpec, "ties": NS.NoteEncodingWithTiesSpec}

# run_length_encoding_test.py:26-38 / note_sequences_test.py:28-40
CODEC = EC.Codec(100, 100, [EC.EventRange("pitch", 0, 127), EC.EventRange("velocity", 0, 127),
                            EC.EventRange("drum", 0, 127), EC.EventRange("program", 0, 127),
                            EC.EventRange("tie", 0, 0)])


def _ns(notes):
    return NS.NoteSequence(notes=[NS.Note(**n) for n in notes])


def test_rle_literals():                           # run_length_encoding_test.py:45-89
    np.testing.assert_array_equal(
        RLE.remove_redundant_state_changes([3, 525, 356, 161, 2, 525, 356, 161, 355, 394], CODEC,
                                           ["velocity", "program"]), [3, 525, 356, 161, 2, 161, 355, 394])
    np.testing.assert_array_equal(RLE.run_length_encode_shifts([1, 1, 1, 161, 1, 1, 1, 162, 1, 1, 1], CODEC),
                                  [3, 161, 6, 162])
    np.testing.assert_array_equal(RLE.run_length_encode_shifts([1] * 202 + [161, 1, 1, 1], CODEC), [100, 100, 2, 161])
    np.testing.assert_array_equal(RLE.run_length_encode_shifts([1, 1, 1, 161, 162, 1, 1, 1], CODEC), [3, 161, 162])
    assert RLE.run_length_encode_shifts([], CODEC).size == 0


def test_encode_and_index_literals():              # note_sequences_test.py:42-100
    ns = _ns([dict(start_time=1.0, end_time=1.1, pitch=61, velocity=100),
              dict(start_time=2.0, end_time=2.1, pitch=62, velocity=100),
              dict(start_time=3.0, end_time=3.1, pitch=63, velocity=100)])
    frame_times = np.arange(0, 4, step=.001)
    times, values = NS.note_sequence_to_onsets(ns)
    ev, st, en, _, _ = RLE.encode_and_index_events(None, times, values, NS.note_event_data_to_events, CODEC, frame_times)
    assert len(st) == len(frame_times) == len(en) and len(ev) == 403
    np.testing.assert_array_equal(ev, [1] * 100 + [162] + [1] * 100 + [163] + [1] * 100 + [164] + [1] * 100)
    assert (st[0], en[0], st[1000], en[1000], st[2000], st[3000], st[-1], en[-1]) == (0, 0, 100, 100, 201, 302, 402, 403)
    # velocity variant (note_sequences_test.py:102-140)
    ns = _ns([dict(start_time=1.0, end_time=3.0, pitch=61, velocity=1),
              dict(start_time=2.0, end_time=4.0, pitch=62, velocity=127)])
    times, values = NS.note_sequence_to_onsets_and_offsets(ns)
    ev, st, en, _, _ = RLE.encode_and_index_events(None, times, values, NS.note_event_data_to_events, CODEC, frame_times)
    assert len(ev) == 408
    np.testing.assert_array_equal(ev[:204], [1] * 100 + [230, 162] + [1] * 100 + [356, 163])


@pytest.mark.parametrize("i", range(len(GOLD["encode_cases"])))
def test_encode_and_index_vs_reference(i):
    c = GOLD["encode_cases"][i]
    codec = V.build_codec(V.VocabularyConfig(num_velocity_bins=c["num_velocity_bins"]))
    ns = _ns(c["notes"])
    fn = {"onsets": NS.note_sequence_to_onsets, "notes": NS.note_sequence_to_onsets_and_offsets,
          "ties": NS.note_sequence_to_onsets_and_offsets_and_programs}[c["mode"]]
    times, values = fn(ns)
    init, enc_fn, state_fn = NS.ENCODING_FNS[SPEC[c["mode"]].name]
    frame_times = np.arange(c["n_frames"]) / 125.0
    ev, st, en, se, si = RLE.encode_and_index_events(init(), times, values, enc_fn, codec, frame_times, state_fn)
    assert [int(x) for x in ev] == c["events"]
    assert [int(x) for x in st] == c["start"] and [int(x) for x in en] == c["end"]
    assert [int(x) for x in se] == c["state_events"] and [int(x) for x in si] == c["state_idx"]


@pytest.mark.parametrize("seed", range(6))
def test_round_trip_notes_tokens_notes(seed):
    """encode (product) -> per-segment targets -> decode (product C++): onsets/offsets survive up to
    the 10 ms grid, pitches/programs/drums exactly."""
    rng = np.random.default_rng(seed)
    codec = V.build_codec(V.VocabularyConfig(num_velocity_bins=1))
    n_seg, T = 3, 256
    total = n_seg * 2.048
    notes = []
    used = set()
    for _ in range(12):
        st = round(float(rng.uniform(0.05, total - 0.5)), 2)
        pitch, prog = int(rng.integers(30, 90)), int(rng.choice([0, 24, 40]))
        if (pitch, prog) in used:
            continue
        used.add((pitch, prog))
        notes.append(dict(start_time=st, end_time=round(st + float(rng.uniform(0.05, 0.45)), 2), pitch=pitch,
                          velocity=100, program=prog, is_drum=False))
    ns = _ns(notes)
    times, values = NS.note_sequence_to_onsets_and_offsets_and_programs(ns)
    frame_times = np.arange(n_seg * T) / 125.0
    init, enc_fn, state_fn = NS.ENCODING_FNS["NoteEncodingWithTiesSpec"]
    enc = RLE.encode_and_index_events(init(), times, values, enc_fn, codec, frame_times, state_fn)
    preds = []
    for s in range(n_seg):
        toks = RLE.segment_targets(*enc, s * T, (s + 1) * T, codec, with_ties=True)
        toks = RLE.remove_redundant_state_changes(toks, codec, ["velocity", "program"])
        st0 = frame_times[s * T]
        preds.append({"est_tokens": toks, "start_time": st0 - st0 % 0.01})
    res = MU.event_predictions_to_ns(preds, codec, NS.NoteEncodingWithTiesSpec)
    assert res["est_invalid_events"] == 0 and res["est_dropped_events"] == 0
    got = sorted((n.pitch, n.program, round(n.start_time, 2), round(n.end_time, 2)) for n in res["est_ns"].notes)
    want = sorted((n["pitch"], n["program"], n["start_time"], max(n["end_time"], n["start_time"] + 0.01)) for n in notes)
    assert [g[:2] for g in got] == [w[:2] for w in want]
    for g, w in zip(got, want):
        assert abs(g[2] - w[2]) <= 0.0101 and abs(g[3] - w[3]) <= 0.0101
