"""Prediction combiner (mirror of mt3/metrics_utils.py:38-146) over libmt3hip.so."""
from __future__ import annotations

import collections
import ctypes as C
from typing import Any, Mapping, Optional, Sequence

import numpy as np

from . import _lib
from . import event_codec
from . import note_sequences


# mt3_note of include/mt3_hip.h as a numpy record: the host stage's output without one Python object per note
NOTE_DTYPE = np.dtype([("start_time", "<f8"), ("end_time", "<f8"), ("pitch", "<i4"), ("velocity", "<i4"),
                       ("program", "<i4"), ("is_drum", "<i4"), ("instrument", "<i4"), ("reserved", "<i4")])
assert NOTE_DTYPE.itemsize == C.sizeof(_lib.NoteStruct)


def _decode(codec, spec_id, flat, offs, start_times, max_times=None):
    """mt3_notes_decode on a flat token array: (notes as a NOTE_DTYPE array, invalid, dropped, total_time)"""
    lib = _lib.load()
    n = len(offs) - 1
    flat = np.ascontiguousarray(flat, np.int32) if n and offs[-1] else np.zeros(1, np.int32)
    offs = np.ascontiguousarray(offs, np.int64)
    st = np.ascontiguousarray(np.asarray(start_times, np.float64).reshape(-1)) if n else np.zeros(1)
    has_p = mt_p = None
    if max_times is not None:
        has = np.array([0 if m is None else 1 for m in max_times], np.int32)
        mts = np.array([0.0 if m is None else float(m) for m in max_times], np.float64)
        has_p, mt_p = has.ctypes.data, mts.ctypes.data
    cap = max(64, int(offs[-1]) + 8)          # a token emits at most one note
    notes = np.empty(cap, NOTE_DTYPE)
    n_notes, inv, drop, total = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(lib.mt3_notes_decode(C.byref(codec.desc), spec_id, n, flat.ctypes.data, offs.ctypes.data,
                                    st.ctypes.data, has_p, mt_p, C.c_void_p(notes.ctypes.data), cap, C.byref(n_notes),
                                    C.byref(inv), C.byref(drop), C.byref(total)))
    return notes[: n_notes.value], inv.value, drop.value, total.value


def decode_token_rows(codec, encoding_spec, rows, start_times, lengths=None):
    """The host stage of a JOB (bench.py, distributed.ShardedTranscriber): `decode_tf`-form token rows int32 [n, L] of the
    consecutive segments of ONE file -> (notes as a NOTE_DTYPE record array, invalid events, dropped events, total_time),
    with no per-row and no per-note Python work (event_predictions_to_ns builds a NoteSequence of Note objects: 2-3 us per
    note under the GIL, which is what bounded rank 0 at N = 8 -- DESIGN.md section 6).  lengths: tokens per row; None = up
    to the first -1 (`_trim_eos`, NB:346-352).  Same combiner rule as event_predictions_to_ns (max_time = the next
    segment's start, mt3/metrics_utils.py:92-116); rows must be in start-time order."""
    rows = np.ascontiguousarray(rows, np.int32)
    if rows.ndim != 2:
        raise ValueError("rows must be [segments, length]")
    if lengths is None:
        eos = rows == -1
        lengths = np.where(eos.any(1), eos.argmax(1), rows.shape[1])
    lengths = np.asarray(lengths, np.int64)
    offs = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lengths, out=offs[1:])
    flat = rows[np.arange(rows.shape[1])[None, :] < lengths[:, None]]
    return _decode(codec, encoding_spec.spec_id, flat, offs, start_times)


def note_sequence_from_records(rec, total_time: float):
    ns = note_sequences.NoteSequence(total_time=total_time)
    Note = note_sequences.Note
    ns.notes = [Note(a, b, p, v, g, bool(d), i) for a, b, p, v, g, d, i, _ in rec.tolist()]
    return ns


def _run(codec, spec_id, tokens_list, start_times, max_times=None):
    n = len(tokens_list)
    toks = [np.asarray(t, np.int32).reshape(-1) for t in tokens_list]
    offs = np.zeros(n + 1, np.int64)
    if n:
        np.cumsum([t.size for t in toks], out=offs[1:])
    flat = np.concatenate(toks) if n and offs[-1] else np.zeros(1, np.int32)
    rec, inv, drop, total = _decode(codec, spec_id, flat, offs, start_times, max_times)
    return note_sequence_from_records(rec, total), inv, drop


def event_predictions_to_ns(predictions: Sequence[Mapping[str, Any]], codec: event_codec.Codec,
                            encoding_spec: note_sequences.NoteEncodingSpecType) -> Mapping[str, Any]:
    """predictions: dicts with 'est_tokens', 'start_time' (and optionally 'raw_inputs')."""
    ns, inv, drop = _run(codec, encoding_spec.spec_id, [p["est_tokens"] for p in predictions],
                         [p["start_time"] for p in predictions])
    order = sorted(range(len(predictions)), key=lambda i: predictions[i]["start_time"])
    raws = [np.asarray(predictions[i].get("raw_inputs", [])) for i in order]
    raws = [r for r in raws if r.size]
    return {
        "raw_inputs": np.concatenate(raws, axis=0) if raws else np.zeros((0,), np.float32),
        "start_times": [predictions[i]["start_time"] for i in order],
        "est_ns": ns,
        "est_invalid_events": inv,
        "est_dropped_events": drop,
    }


def decode_events_single(tokens, start_time, max_time: Optional[float], codec: event_codec.Codec,
                         encoding_spec: note_sequences.NoteEncodingSpecType):
    """One call of run_length_encoding.decode_events on a fresh state + flush."""
    return _run(codec, encoding_spec.spec_id, [tokens], [start_time], [max_time])


def combine_predictions_by_id(predictions, combine_predictions_fn):
    by_id = collections.defaultdict(list)
    for p in predictions:
        by_id[p["unique_id"]].append(p)
    return {k: combine_predictions_fn(v) for k, v in by_id.items()}
