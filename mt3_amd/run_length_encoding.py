"""Run-length encoding of event streams (mirror of mt3/run_length_encoding.py).

Decode (`decode_events`) runs in libmt3hip.so (csrc/symbolic.cpp).  The encode side
(`encode_and_index_events`, `run_length_encode_shifts`,
`remove_redundant_state_changes`, reference lines 63-295) is host integer work, done
here with numpy instead of the reference's per-token Python/tf.autograph loops:
events are grouped by their quantised step, shift runs and per-frame indices come
from prefix sums and `searchsorted`.  Pinned bit-exactly by the reference's own
literals and by goldens produced with the reference's real modules
(tests/test_encoding.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from . import event_codec
from . import metrics_utils
from . import note_sequences

Event = event_codec.Event


def decode_events(tokens, start_time, max_time, codec: event_codec.Codec,
                  encoding_spec: note_sequences.NoteEncodingSpecType):
    """One segment on a fresh decoding state, flushed: (NoteSequence, invalid, dropped).
    (run_length_encoding.py:371-423 + the spec's flush.)"""
    return metrics_utils.decode_events_single(tokens, start_time, max_time, codec, encoding_spec)


def _first_step_after(frame_times: np.ndarray, sps: float) -> np.ndarray:
    """k[f] = the smallest integer k >= 1 with frame_times[f] < k / sps, using exactly that float
    comparison (the reference's `frame_times[i] < cur_step / codec.steps_per_second`)."""
    ft = np.asarray(frame_times, np.float64)
    k = np.floor(ft * sps).astype(np.int64) + 1
    k = np.maximum(k, 1)
    for _ in range(3):                                   # repair float rounding either way
        k = np.where(ft < (k - 1) / sps, k - 1, k)
        k = np.where(ft < k / sps, k, k + 1)
        k = np.maximum(k, 1)
    return k


def encode_and_index_events(state, event_times: Sequence[float], event_values: Sequence,
                            encode_event_fn: Callable, codec: event_codec.Codec, frame_times: Sequence[float],
                            encoding_state_to_events_fn: Optional[Callable] = None
                            ) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Timed events -> unit-shift token stream + per-audio-frame start/end indices (+ "state" events
    captured before every event, for tie sections).  Returns (events, event_start_indices,
    event_end_indices, state_events, state_event_indices)."""
    sps = codec.steps_per_second
    frame_times = np.asarray(frame_times, np.float64)
    order = np.argsort(np.asarray(event_times, np.float64), kind="stable")
    steps = [round(event_times[i] * sps) for i in order]           # Python round: half to even
    shift_tok = codec.encode_event(Event("shift", 1))

    # tokens of every event, in order; state tokens captured BEFORE the event mutates the state
    ev_tokens, st_tokens = [], []
    for i in order:
        st_tokens.append([codec.encode_event(e) for e in encoding_state_to_events_fn(state)]
                         if encoding_state_to_events_fn else [])
        ev_tokens.append([codec.encode_event(e) for e in encode_event_fn(state, event_values[i], codec)])

    last_event_step = 0
    for s in steps:                                                 # cur_step only ever moves forward
        last_event_step = max(last_event_step, s)
    # trailing shifts: until cur_step / sps > frame_times[-1]
    n_shift = last_event_step
    while n_shift / sps <= frame_times[-1]:
        n_shift += 1

    # assemble: before an event at step s (s > current) come the missing unit shifts
    events: list = []
    state_events: list = []
    after_shift = np.zeros(n_shift + 1, np.int64)          # event index right after the j-th shift
    state_after_shift = np.zeros(n_shift + 1, np.int64)    # state-event index at that moment
    cur = 0
    for s, toks, stoks in zip(steps, ev_tokens, st_tokens):
        while cur < s:
            events.append(shift_tok)
            cur += 1
            after_shift[cur] = len(events)
            state_after_shift[cur] = len(state_events)
        state_events.extend(stoks)
        events.extend(toks)
    frozen_state_idx = state_after_shift[cur]               # trailing shifts do not refresh it
    while cur < n_shift:
        events.append(shift_tok)
        cur += 1
        after_shift[cur] = len(events)
        state_after_shift[cur] = frozen_state_idx

    k = _first_step_after(frame_times, sps)                 # frame f is indexed when cur_step reaches k[f]
    start = after_shift[k - 1]
    state_idx = state_after_shift[k - 1]
    end = np.concatenate([start[1:], [len(events)]])
    return (np.array(events), start, end, np.array(state_events), state_idx)


def run_length_encode_shifts(events: Sequence[int], codec: event_codec.Codec) -> np.ndarray:
    """Unit shifts -> absolute "steps since segment start" tokens emitted before each non-shift
    event (values above max_shift_steps split), trailing shifts dropped
    (run_length_encode_shifts_fn, run_length_encoding.py:242-295)."""
    ev = np.asarray(events, np.int64)
    is_shift = (ev >= 0) & (ev <= codec.max_shift_steps)
    total = np.cumsum(is_shift)                              # unit shifts seen up to each position
    out = []
    seen = 0                                                 # value of `total` at the previous non-shift event
    for i in np.flatnonzero(~is_shift):
        if total[i] > seen:                                  # shifts since the last event: emit the absolute count
            rem = int(total[i])
            while rem > 0:
                o = min(codec.max_shift_steps, rem)
                out.append(o)
                rem -= o
            seen = int(total[i])
        out.append(int(ev[i]))
    return np.array(out, np.int32)


def remove_redundant_state_changes(events: Sequence[int], codec: event_codec.Codec,
                                   state_change_event_types: Sequence[str] = ()) -> np.ndarray:
    """Drop a state-change token (e.g. velocity, program) equal to the current value of its type
    (remove_redundant_state_changes_fn, run_length_encoding.py:194-239)."""
    ranges = [codec.event_type_range(t) for t in state_change_event_types]
    current = [0] * len(ranges)
    out = []
    for e in events:
        e = int(e)
        redundant = False
        for i, (lo, hi) in enumerate(ranges):
            if lo <= e <= hi:
                redundant = redundant or current[i] == e
                current[i] = e
        if not redundant:
            out.append(e)
    return np.array(out, np.int32)


def segment_targets(events, event_start_indices, event_end_indices, state_events, state_event_indices,
                    frame_lo: int, frame_hi: int, codec: event_codec.Codec, with_ties: bool) -> np.ndarray:
    """Targets of the audio segment covering frames [frame_lo, frame_hi): the tie-section state
    tokens (up to and including the first `tie`), then the segment's events, run-length encoded
    (extract_target_sequence_with_indices + run_length_encode_shifts, run_length_encoding.py:170-191)."""
    seg = np.asarray(events)[event_start_indices[frame_lo]: event_end_indices[frame_hi - 1]]
    toks = run_length_encode_shifts(seg, codec)
    if with_ties:
        st = [int(x) for x in np.asarray(state_events)[state_event_indices[frame_lo]:]]
        tie = codec.encode_event(Event("tie", 0))
        st = st[: st.index(tie) + 1] if tie in st else [tie]
        toks = np.concatenate([np.array(st, np.int32), toks])
    return toks.astype(np.int32)
