"""`write_inferences_to_file` (the second half of the drop-in boundary) against the reference's behaviour,
mt3/inference.py:34-138: the three ValueErrors (:57-63), grouping by `unique_id` with segments in ANY order
(metrics_utils.combine_predictions_by_id sorts by start_time), start-time flooring from `input_times[0]` (:80-82),
`decode_tf` + `trim_eos` on raw model ids (:78), the JSON-lines schema (:120-138).  Expected notes are the
reference-run goldens (tests/golden/symbolic_golden.json: produced by the reference's REAL run_length_encoding /
note_sequences / metrics_utils, tests/golden/make_symbolic_golden.py).

Host-only: the symbolic stage of libmt3hip.so is C++ and `decode_tf` remaps host arrays on the host, so this
runs without a GPU.
"""
import json
import os

import numpy as np
import pytest

from mt3_amd import inference, vocabularies as V

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "symbolic_golden.json")) as f:
    GOLD = json.load(f)

L = 1024


def _raw_time(start):
    """a frame time whose flooring to the 10 ms grid gives the golden's start_time (segments start at k * 2.048 s
    or k * 4.096 s; the golden stores the floored value)"""
    for seg in (2.048, 4.096):
        k = round(start / seg)
        for kk in (k - 1, k, k + 1):
            t = kk * seg
            if kk >= 0 and t - t % 0.01 == start:
                return t
    raise AssertionError(start)


def _ids_row(tokens, rng):
    """model ids of one segment: token + 3, EOS, then whatever the decoder kept emitting (must be ignored)"""
    row = np.zeros(L, np.int32)
    n = min(len(tokens), L - 1)
    row[:n] = np.asarray(tokens[:n], np.int32) + 3
    row[n] = 1
    row[n + 1:] = rng.integers(0, 1491, size=L - n - 1)
    return row


def _track(case, uid, rng, shuffle):
    exs, rows = [], []
    for s in case["segments"]:
        t0 = _raw_time(s["start_time"])
        exs.append({"input_times": np.array([t0, t0 + 0.008]), "unique_id": np.array([uid.encode()]),
                    "raw_inputs": np.zeros((0,), np.float32)})
        rows.append(_ids_row(s["tokens"], rng))
    order = rng.permutation(len(exs)) if shuffle else np.arange(len(exs))
    return [exs[i] for i in order], [rows[i] for i in order]


def _cases(mode, nvb, n):
    # tokens must survive the id round trip (token + 3 -> decode_tf): regular codec classes only
    ncls = V.build_codec(V.VocabularyConfig(num_velocity_bins=nvb)).num_classes
    out = [c for c in GOLD["decode_cases"] if c["mode"] == mode and c["num_velocity_bins"] == nvb
           and all(len(s["tokens"]) < L and all(0 <= t < ncls for t in s["tokens"]) for s in c["segments"])]
    assert len(out) >= n, (mode, nvb, len(out))
    return out[:n]


@pytest.mark.parametrize("mode,nvb,onsets_only,use_ties", [("ties", 1, False, True), ("notes", 127, False, False),
                                                           ("onsets", 127, True, False)])
def test_jsonl_matches_reference_goldens(tmp_path, mode, nvb, onsets_only, use_ties):
    present = {(c["mode"], c["num_velocity_bins"]) for c in GOLD["decode_cases"]}
    if (mode, nvb) not in present:
        nvb = next(v for m, v in sorted(present) if m == mode)
    rng = np.random.default_rng(7)
    cases = _cases(mode, nvb, 3)
    ds, infs, want = [], [], {}
    for i, c in enumerate(cases):
        uid = "track_%c" % "cab"[i]                      # ids out of order: output lines are sorted by id
        e, r = _track(c, uid, rng, shuffle=True)
        ds += e
        infs += r
        want[uid] = c["notes"]
    # interleave the tracks' segments as a batched infer job would
    order = rng.permutation(len(ds))
    ds, infs = [ds[i] for i in order], [infs[i] for i in order]
    cfg = V.VocabularyConfig(num_velocity_bins=nvb)
    vocab = V.vocabulary_from_codec(V.build_codec(cfg))
    path = str(tmp_path / "inferences.jsonl")
    inference.write_inferences_to_file(path, infs, ds, mode="predict", vocabulary=vocab, vocab_config=cfg,
                                       onsets_only=onsets_only, use_ties=use_ties)
    lines = [json.loads(l) for l in open(path)]
    assert [l["id"] for l in lines] == sorted(want)
    for l in lines:
        assert set(l) == {"id", "est_notes"}
        got = [[n["start_time"], n["end_time"], n["pitch"], n["velocity"], n["program"], n["is_drum"]]
               for n in l["est_notes"]]
        assert got == [n[:6] for n in want[l["id"]]]
        for n in l["est_notes"]:
            assert list(n) == ["start_time", "end_time", "pitch", "velocity", "program", "is_drum"]
            assert isinstance(n["is_drum"], bool) and isinstance(n["pitch"], int) and isinstance(n["start_time"], float)


def test_errors_match_the_reference(tmp_path):
    cfg = V.VocabularyConfig(num_velocity_bins=1)
    vocab = V.vocabulary_from_codec(V.build_codec(cfg))
    p = str(tmp_path / "x.jsonl")
    kw = dict(vocab_config=cfg, onsets_only=False, use_ties=True)
    with pytest.raises(ValueError, match="score"):                       # inference.py:57-58
        inference.write_inferences_to_file(p, [], [], mode="score", vocabulary=vocab, **kw)
    with pytest.raises(ValueError, match="vocabulary"):                  # :59-60
        inference.write_inferences_to_file(p, [], [], mode="predict", vocabulary=None, **kw)
    with pytest.raises(ValueError, match="ties not compatible"):         # :62-63
        inference.write_inferences_to_file(p, [], [], mode="predict", vocabulary=vocab, vocab_config=cfg,
                                           onsets_only=True, use_ties=True)
    with pytest.raises(ValueError):                                      # gin.REQUIRED left unbound
        inference.write_inferences_to_file(p, [], [], mode="predict", vocabulary=vocab)
    # `score` is checked before the vocabulary, as in the reference
    with pytest.raises(ValueError, match="score"):
        inference.write_inferences_to_file(p, [], [], mode="score", vocabulary=None, **kw)
    assert not os.path.exists(p)
    # no examples: an empty file, no error
    inference.write_inferences_to_file(p, [], [], mode="predict", vocabulary=vocab, **kw)
    assert open(p).read() == ""


def test_ids_without_eos_invalid_ids_and_scalar_fields(tmp_path):
    """A row that never emits EOS is used whole; special / sentinel ids become invalid events, not errors
    (reference: malformed model output is counted, never raised); plain-scalar `unique_id` / `input_times` work."""
    cfg = V.VocabularyConfig(num_velocity_bins=1)
    codec = V.build_codec(cfg)
    vocab = V.vocabulary_from_codec(codec)
    c = _cases("ties", 1, 1)[0]
    rng = np.random.default_rng(3)
    ds, infs = [], []
    for s in c["segments"]:
        row = np.zeros(L, np.int32)
        toks = np.asarray(s["tokens"], np.int32) + 3
        row[: len(toks)] = toks
        row[len(toks)] = 1
        ds.append({"input_times": _raw_time(s["start_time"]), "unique_id": "solo"})
        infs.append(row)
    p = str(tmp_path / "solo.jsonl")
    inference.write_inferences_to_file(p, infs, ds, mode="predict", vocabulary=vocab, vocab_config=cfg,
                                       onsets_only=False, use_ties=True)
    (line,) = [json.loads(l) for l in open(p)]
    assert line["id"] == "solo" and len(line["est_notes"]) == len(c["notes"])
    # a second file whose only row has no EOS and holds pad / unk / sentinel ids
    row = rng.integers(1389 + 3, 1536, size=L).astype(np.int32)
    row[::5] = 2
    p2 = str(tmp_path / "junk.jsonl")
    inference.write_inferences_to_file(p2, [row], [{"input_times": [0.0], "unique_id": [b"junk"]}], mode="predict",
                                       vocabulary=vocab, vocab_config=cfg, onsets_only=False, use_ties=True)
    (line,) = [json.loads(l) for l in open(p2)]
    assert line == {"id": "junk", "est_notes": []}


def test_decode_tf_host_path_matches_the_reference_literals():
    """vocabularies_test.py:47-83 literals on the host path (no GPU), incl. an empty row."""
    vocab = V.GenericTokenVocabulary(10, extra_ids=4)
    ids = np.array([[3, 4, 5, 1, 7, 1], [12, 13, 2, 0, 1, 3]], np.int32)
    assert vocab.decode_tf(ids).tolist() == [[0, 1, 2, -1, -1, -1], [9, -2, -2, -2, -1, -1]]
    assert vocab.decode_tf(np.zeros((2, 0), np.int32)).shape == (2, 0)
    assert vocab.decode([3, 4, 1, 5]) == [0, 1, -1]


def _proto_note_sequence(ident: str, filename: str = "a.mid") -> bytes:
    """a serialized note_seq.NoteSequence written by hand from the protobuf wire format: field 2 (filename) FIRST,
    a varint field (ticks_per_quarter = 4: 220) and a fixed64 one in between, then field 1 (id) -- order is free on
    the wire, the reader must skip what it does not want"""
    def ld(field, payload):
        assert len(payload) < 128
        return bytes([(field << 3) | 2, len(payload)]) + payload
    return (ld(2, filename.encode()) + bytes([(4 << 3) | 0, 0xDC, 0x01]) +
            bytes([(6 << 3) | 1]) + b"\0" * 8 + ld(1, ident.encode()))


def test_id_comes_from_the_reference_note_sequence(tmp_path):
    """mt3/inference.py:83-86,104-108,133: when the task dataset carries the ground-truth `sequence` (first segment
    of each track; later ones are empty), the line's "id" is that NoteSequence's `id`, not the `unique_id`; lines are
    ordered by `unique_id`; a track without any sequence trips the reference's assertion (:114)."""
    cfg = V.VocabularyConfig(num_velocity_bins=1)
    vocab = V.vocabulary_from_codec(V.build_codec(cfg))
    rng = np.random.default_rng(11)
    ds, infs = [], []
    for uid, ref_id, case in (("u2", "/id/slakh/Track00017", _cases("ties", 1, 2)[0]),
                              ("u1", "/id/maestro/zzz", _cases("ties", 1, 2)[1])):
        e, r = _track(case, uid, rng, shuffle=False)
        for k, ex in enumerate(e):
            ex["sequence"] = np.array([_proto_note_sequence(ref_id) if k == 0 else b""], dtype=object)
        ds += e
        infs += r
    assert inference.note_sequence_id(_proto_note_sequence("abc")) == "abc"
    assert inference.note_sequence_id(b"") == "" and inference.note_sequence_id("plain") == "plain"
    p = str(tmp_path / "ids.jsonl")
    inference.write_inferences_to_file(p, infs, ds, mode="predict", vocabulary=vocab, vocab_config=cfg,
                                       onsets_only=False, use_ties=True)
    assert [json.loads(l)["id"] for l in open(p)] == ["/id/maestro/zzz", "/id/slakh/Track00017"]      # sorted by unique_id
    for ex in ds:
        if ex["unique_id"][0] == b"u1":
            ex["sequence"] = np.array([b""], dtype=object)
    with pytest.raises(AssertionError) as ei:              # the reference's exception type ...
        inference.write_inferences_to_file(p, infs, ds, mode="predict", vocabulary=vocab, vocab_config=cfg,
                                           onsets_only=False, use_ties=True)
    assert isinstance(ei.value, ValueError) and "u1" in str(ei.value)      # ... raised, not asserted: survives python -O
    with pytest.raises(ValueError):                        # truncated proto bytes: an error, not an IndexError
        inference.note_sequence_id(_proto_note_sequence("abcdef")[:4])
