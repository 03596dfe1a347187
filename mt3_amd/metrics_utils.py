"""Prediction combiner (mirror of mt3/metrics_utils.py:38-146) over libmt3hip.so."""
from __future__ import annotations

import collections
import ctypes as C
from typing import Any, Mapping, Optional, Sequence

import numpy as np

from . import _lib
from . import event_codec
from . import note_sequences


def _run(codec, spec_id, tokens_list, start_times, max_times=None):
    lib = _lib.load()
    n = len(tokens_list)
    toks = [np.ascontiguousarray(np.asarray(t).reshape(-1), dtype=np.int32) for t in tokens_list]
    offs = np.zeros(n + 1, np.int64)
    for i, t in enumerate(toks):
        offs[i + 1] = offs[i] + t.size
    flat = np.concatenate(toks) if n and offs[-1] else np.zeros(1, np.int32)
    st = np.ascontiguousarray(np.asarray(start_times, np.float64).reshape(-1)) if n else np.zeros(1)
    has_p = mt_p = None
    if max_times is not None:
        has = np.array([0 if m is None else 1 for m in max_times], np.int32)
        mts = np.array([0.0 if m is None else float(m) for m in max_times], np.float64)
        has_p, mt_p = has.ctypes.data, mts.ctypes.data
    cap = max(64, int(offs[-1]) + 8)          # a token emits at most one note
    notes = (_lib.NoteStruct * cap)()
    n_notes, inv, drop, total = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(lib.mt3_notes_decode(C.byref(codec.desc), spec_id, n, flat.ctypes.data, offs.ctypes.data,
                                    st.ctypes.data, has_p, mt_p, C.cast(notes, C.c_void_p), cap, C.byref(n_notes),
                                    C.byref(inv), C.byref(drop), C.byref(total)))
    ns = note_sequences.NoteSequence(total_time=total.value)
    for i in range(n_notes.value):
        s = notes[i]
        ns.notes.append(note_sequences.Note(s.start_time, s.end_time, s.pitch, s.velocity, s.program,
                                            bool(s.is_drum), s.instrument))
    return ns, inv.value, drop.value


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
