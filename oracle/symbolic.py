"""Oracle (CPU, pure Python) for the integer / symbolic stage of the MT3 path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Restates, citing reference
file:line, the part of magenta/mt3 that turns model ids into notes:

  ids -> tokens                mt3/vocabularies.py:148-277 (GenericTokenVocabulary)
  token -> (type, value)       mt3/event_codec.py:34-112 (Codec)
  codec layout                 mt3/vocabularies.py:119-140 (build_codec)
  run-length decode            mt3/run_length_encoding.py:371-423 (decode_events)
  note state machine           mt3/note_sequences.py:262-446
  segment combiner             mt3/metrics_utils.py:59-146
  trim / start-time flooring   notebook InferenceModel.postprocess/_trim_eos
                               (mt3/colab/music_transcription_with_transformers.ipynb,
                               cell "Imports and Definitions"), mt3/tasks.py:58-63

Written from the reference's behaviour, not its text: a table-driven codec, a
decoder *object* instead of free functions over a dataclass, and a plain
``Note``/``NoteSeq`` pair standing in for the ``note_seq.NoteSequence`` proto.
Pinned by tests/test_oracle_symbolic.py (reference unit-test literals) and
tests/golden/symbolic_golden.json (outputs of the real reference modules).
"""
from __future__ import annotations

import bisect
import dataclasses
import math
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

# note_seq constants the reference reads (vocabularies.py:67,74,122-133) -- values
# pinned indirectly by the reference's literals (pitch 60 -> 161 with 101 shifts, ...).
MIN_MIDI_PITCH, MAX_MIDI_PITCH = 0, 127
MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM = 0, 127
MAX_MIDI_VELOCITY = 127

DECODED_EOS_ID = -1       # vocabularies.py:29
DECODED_INVALID_ID = -2   # vocabularies.py:30
DEFAULT_EXTRA_IDS = 100   # t5.data.DEFAULT_EXTRA_IDS [third-party; from memory]

DEFAULT_VELOCITY = 100        # note_sequences.py:28
DEFAULT_NOTE_DURATION = 0.01  # note_sequences.py:29
MIN_NOTE_DURATION = 0.01      # note_sequences.py:32


# --------------------------------------------------------------------------- codec
@dataclasses.dataclass(frozen=True)
class Event:
    type: str
    value: int


class Codec:
    """Flat-index <-> (type, value) bijection; `shift` always occupies ids 0..max.

    Follows event_codec.py:34-112.  Implemented as a cumulative-offset table
    with a bisect lookup instead of the reference's linear range walk.
    """

    def __init__(self, max_shift_steps: int, steps_per_second: float,
                 ranges: Sequence[Tuple[str, int, int]]):
        self.steps_per_second = steps_per_second
        self.ranges: List[Tuple[str, int, int]] = [("shift", 0, int(max_shift_steps))]
        self.ranges += [(str(t), int(lo), int(hi)) for t, lo, hi in ranges]
        names = [r[0] for r in self.ranges]
        if len(set(names)) != len(names):
            raise AssertionError("event types must be unique")
        self._starts: List[int] = []
        off = 0
        for _, lo, hi in self.ranges:
            self._starts.append(off)
            off += hi - lo + 1
        self._total = off

    @property
    def num_classes(self) -> int:
        return self._total

    @property
    def max_shift_steps(self) -> int:
        return self.ranges[0][2]

    def is_shift_event_index(self, index: int) -> bool:
        return 0 <= index <= self.ranges[0][2]

    def event_type_range(self, event_type: str) -> Tuple[int, int]:
        for (t, lo, hi), start in zip(self.ranges, self._starts):
            if t == event_type:
                return start, start + (hi - lo)
        raise ValueError(f"Unknown event type: {event_type}")

    def encode_event(self, event: Event) -> int:
        for (t, lo, hi), start in zip(self.ranges, self._starts):
            if t == event.type:
                if not lo <= event.value <= hi:
                    raise ValueError(
                        f"Event value {event.value} is not within valid range "
                        f"[{lo}, {hi}] for type {event.type}")
                return start + event.value - lo
        raise ValueError(f"Unknown event type: {event.type}")

    def decode_event_index(self, index: int) -> Event:
        index = int(index)
        if index < 0 or index >= self._total:
            raise ValueError(f"Unknown event index: {index}")
        k = bisect.bisect_right(self._starts, index) - 1
        t, lo, _ = self.ranges[k]
        return Event(t, lo + index - self._starts[k])


@dataclasses.dataclass
class VocabularyConfig:
    """vocabularies.py:38-54."""
    steps_per_second: int = 100
    max_shift_seconds: int = 10
    num_velocity_bins: int = 127


def build_codec(cfg: VocabularyConfig) -> Codec:
    """vocabularies.py:119-140: shift | pitch | velocity(0..bins) | tie | program | drum."""
    return Codec(
        max_shift_steps=cfg.steps_per_second * cfg.max_shift_seconds,
        steps_per_second=cfg.steps_per_second,
        ranges=[
            ("pitch", MIN_MIDI_PITCH, MAX_MIDI_PITCH),
            ("velocity", 0, cfg.num_velocity_bins),
            ("tie", 0, 0),
            ("program", MIN_MIDI_PROGRAM, MAX_MIDI_PROGRAM),
            ("drum", MIN_MIDI_PITCH, MAX_MIDI_PITCH),
        ])


def num_velocity_bins_from_codec(codec: Codec) -> int:
    lo, hi = codec.event_type_range("velocity")   # vocabularies.py:57-60
    return hi - lo


def velocity_to_bin(velocity: int, num_velocity_bins: int) -> int:
    if velocity == 0:                               # vocabularies.py:63-67
        return 0
    return math.ceil(num_velocity_bins * velocity / MAX_MIDI_VELOCITY)


def bin_to_velocity(velocity_bin: int, num_velocity_bins: int) -> int:
    if velocity_bin == 0:                           # vocabularies.py:70-74
        return 0
    return int(MAX_MIDI_VELOCITY * velocity_bin / num_velocity_bins)


# ---------------------------------------------------------------------- vocabulary
class GenericTokenVocabulary:
    """ids <-> tokens with PAD=0, EOS=1, UNK=2 in front (vocabularies.py:148-277).

    The seqio.Vocabulary base-class behaviour that the reference inherits
    (unk-replacement of ids >= base size, EOS truncation in `decode`, pad-fill
    after EOS in `decode_tf`) is third-party [from memory]; the end-to-end
    results are pinned by mt3/vocabularies_test.py:47-83.
    """
    NUM_SPECIAL = 3

    def __init__(self, regular_ids: int, extra_ids: int = 0):
        self._num_regular_tokens = int(regular_ids)
        self.extra_ids = int(extra_ids)

    pad_id, eos_id, unk_id = 0, 1, 2

    @property
    def _base_vocab_size(self) -> int:
        return self.NUM_SPECIAL + self._num_regular_tokens

    @property
    def vocab_size(self) -> int:
        return self._base_vocab_size + self.extra_ids

    def encode(self, token_ids: Sequence[int]) -> List[int]:
        out = []
        for t in token_ids:
            if not 0 <= t < self._num_regular_tokens:
                raise ValueError(
                    f"token_id {t} does not fall within valid range of "
                    f"[0, {self._num_regular_tokens})")
            out.append(int(t) + self.NUM_SPECIAL)
        return out

    def _map_one(self, i: int) -> int:
        if i == self.eos_id:
            return DECODED_EOS_ID
        if i < self.NUM_SPECIAL or i >= self._base_vocab_size:
            return DECODED_INVALID_ID
        return i - self.NUM_SPECIAL

    def decode(self, ids: Sequence[int]) -> List[int]:
        """Python path: truncate after the first EOS (vocabularies_test.py:73-78)."""
        ids = [int(i) for i in ids]
        ids = [self.unk_id if i >= self._base_vocab_size else i for i in ids]
        if self.eos_id in ids:
            ids = ids[: ids.index(self.eos_id) + 1]
        return [self._map_one(i) for i in ids]

    def decode_tf(self, ids) -> np.ndarray:
        """TF path used by predict_tokens: -1 from the first EOS to the end of the
        row, id-3 for regular ids, -2 otherwise (vocabularies.py:241-271)."""
        ids = np.asarray(ids)
        eos_and_after = np.cumsum(ids == self.eos_id, axis=-1) > 0
        regular = (ids >= self.NUM_SPECIAL) & (ids < self._base_vocab_size)
        out = np.where(regular, ids - self.NUM_SPECIAL, DECODED_INVALID_ID)
        out = np.where(eos_and_after, DECODED_EOS_ID, out)
        return out.astype(ids.dtype if np.issubdtype(ids.dtype, np.integer) else np.int32)

    def __eq__(self, other):
        return (self.extra_ids == other.extra_ids and
                self._num_regular_tokens == other._num_regular_tokens)


def vocabulary_from_codec(codec: Codec) -> GenericTokenVocabulary:
    return GenericTokenVocabulary(codec.num_classes, extra_ids=DEFAULT_EXTRA_IDS)


def num_embeddings(vocabulary: GenericTokenVocabulary) -> int:
    return 128 * math.ceil(vocabulary.vocab_size / 128)   # vocabularies.py:280-282


def trim_eos(tokens) -> np.ndarray:
    """Cut at the first -1 (notebook `_trim_eos`; tasks.py:58-63)."""
    tokens = np.array(tokens, np.int32)
    hits = np.nonzero(tokens == DECODED_EOS_ID)[0]
    return tokens[: hits[0]] if hits.size else tokens


def floor_start_time(t: float, steps_per_second: float) -> float:
    """`start_time -= start_time % (1 / steps_per_second)` in float64 (notebook
    postprocess; inference.py:79-81)."""
    t = float(t)
    return t - t % (1 / steps_per_second)


# ------------------------------------------------------------------ note container
@dataclasses.dataclass
class Note:
    start_time: float = 0.0
    end_time: float = 0.0
    pitch: int = 0
    velocity: int = 0
    program: int = 0
    is_drum: bool = False
    instrument: int = 0


@dataclasses.dataclass
class NoteSeq:
    """Stand-in for note_seq.NoteSequence (fields the path touches only)."""
    notes: List[Note] = dataclasses.field(default_factory=list)
    total_time: float = 0.0
    ticks_per_quarter: int = 220

    def as_tuples(self):
        return [(n.start_time, n.end_time, n.pitch, n.velocity, n.program,
                 bool(n.is_drum), n.instrument) for n in self.notes]


def assign_instruments(ns: NoteSeq) -> None:
    """Instrument = order of first appearance of the program, skipping 9; drums
    are 9 (note_sequences.py:72-84)."""
    seen: Dict[int, int] = {}
    for n in ns.notes:
        if n.is_drum:
            n.instrument = 9
        elif n.program in seen:
            n.instrument = seen[n.program]
        else:
            k = len(seen)
            n.instrument = k if k < 9 else k + 1
            seen[n.program] = n.instrument


# -------------------------------------------------------------- note state machine
class NoteDecoder:
    """The reference's NoteDecodingState + decode fns (note_sequences.py:262-408)
    as one object.  `mode`: 'onsets' (NoteOnsetEncodingSpec), 'notes'
    (NoteEncodingSpec) or 'ties' (NoteEncodingWithTiesSpec) -- note_sequences.py:416-446.
    """

    def __init__(self, mode: str):
        if mode not in ("onsets", "notes", "ties"):
            raise ValueError(mode)
        self.mode = mode
        self.current_time = 0.0
        self.current_velocity = DEFAULT_VELOCITY
        self.current_program = 0
        # insertion-ordered: (pitch, program) -> (onset_time, onset_velocity)
        self.active: Dict[Tuple[int, int], Tuple[float, int]] = {}
        self.tied: set = set()
        self.in_tie_section = False
        self.ns = NoteSeq()

    # -- helpers
    def _emit(self, start, end, pitch, velocity, program=0, is_drum=False):
        end = max(end, start + MIN_NOTE_DURATION)            # note_sequences.py:306
        self.ns.notes.append(Note(start, end, pitch, velocity, program, is_drum))
        self.ns.total_time = max(self.ns.total_time, end)

    # -- spec hooks
    def begin_segment(self):
        if self.mode == "ties":                               # note_sequences.py:390-393
            self.tied = set()
            self.in_tie_section = True

    def consume(self, time: float, ev: Event, codec: Codec):
        if self.mode == "onsets":                             # note_sequences.py:284-298
            if ev.type != "pitch":
                raise ValueError(f"unexpected event type: {ev.type}")
            self.ns.notes.append(Note(time, time + DEFAULT_NOTE_DURATION, ev.value,
                                      DEFAULT_VELOCITY))
            self.ns.total_time = max(self.ns.total_time, time + DEFAULT_NOTE_DURATION)
            return
        # note_sequences.py:313-387
        if time < self.current_time:
            raise ValueError("event time < current time")
        self.current_time = time
        kind = ev.type
        if kind == "pitch":
            key = (ev.value, self.current_program)
            if self.in_tie_section:
                if key not in self.active:
                    raise ValueError("inactive pitch/program in tie section")
                if key in self.tied:
                    raise ValueError("pitch/program is already tied")
                self.tied.add(key)
            elif self.current_velocity == 0:
                if key not in self.active:
                    raise ValueError("note-off for inactive pitch/program")
                t0, v0 = self.active.pop(key)
                self._emit(t0, time, ev.value, v0, self.current_program)
            else:
                if key in self.active:
                    t0, v0 = self.active.pop(key)
                    self._emit(t0, time, ev.value, v0, self.current_program)
                self.active[key] = (time, self.current_velocity)
        elif kind == "drum":
            if self.current_velocity == 0:
                raise ValueError("velocity cannot be zero for drum event")
            self._emit(time, time + DEFAULT_NOTE_DURATION, ev.value,
                       self.current_velocity, 0, True)
        elif kind == "velocity":
            self.current_velocity = bin_to_velocity(
                ev.value, num_velocity_bins_from_codec(codec))
        elif kind == "program":
            self.current_program = ev.value
        elif kind == "tie":
            if not self.in_tie_section:
                raise ValueError("tie section end event when not in tie section")
            for key in list(self.active.keys()):
                if key not in self.tied:
                    t0, v0 = self.active.pop(key)
                    self._emit(t0, self.current_time, key[0], v0, key[1])
            self.in_tie_section = False
        else:
            raise ValueError(f"unexpected event type: {kind}")

    def flush(self) -> NoteSeq:
        if self.mode == "onsets":
            return self.ns
        for t0, _ in self.active.values():                    # note_sequences.py:399-401
            self.current_time = max(self.current_time, t0 + MIN_NOTE_DURATION)
        for key in list(self.active.keys()):
            t0, v0 = self.active.pop(key)
            self._emit(t0, self.current_time, key[0], v0, key[1])
        assign_instruments(self.ns)
        return self.ns


def decode_events(state: NoteDecoder, tokens: Iterable[int], start_time, max_time,
                  codec: Codec) -> Tuple[int, int]:
    """run_length_encoding.py:371-423.  Shift values are absolute steps since the
    segment start, accumulated across a run of shift tokens; any other valid
    token resets the run.  `max_time` is tested by truthiness and strictly."""
    tokens = list(tokens)
    invalid = dropped = 0
    steps = 0
    now = start_time
    for pos, tok in enumerate(tokens):
        try:
            ev = codec.decode_event_index(tok)
        except ValueError:
            invalid += 1
            continue
        if ev.type == "shift":
            steps += ev.value
            now = start_time + steps / codec.steps_per_second
            if max_time and now > max_time:
                dropped = len(tokens) - pos
                break
        else:
            steps = 0
            try:
                state.consume(now, ev, codec)
            except ValueError:
                invalid += 1
    return invalid, dropped


def event_predictions_to_ns(predictions: Sequence[dict], codec: Codec, mode: str) -> dict:
    """metrics_utils.py:59-146: stable-sort segments by start_time, walk them with
    ONE decoder, clip each segment at the next one's start."""
    order = sorted(range(len(predictions)), key=lambda i: predictions[i]["start_time"])
    preds = [predictions[i] for i in order]
    dec = NoteDecoder(mode)
    invalid = dropped = 0
    for k, p in enumerate(preds):
        dec.begin_segment()
        limit = preds[k + 1]["start_time"] if k + 1 < len(preds) else None
        a, b = decode_events(dec, p["est_tokens"], p["start_time"], limit, codec)
        invalid += a
        dropped += b
    ns = dec.flush()
    return {
        "start_times": [p["start_time"] for p in preds],
        "est_ns": ns,
        "est_invalid_events": invalid,
        "est_dropped_events": dropped,
    }


# --------------------------------------------------------- audio framing (host int)
def audio_to_frames(audio: np.ndarray, hop: int = 128, frames_per_second: float = 125.0):
    """Notebook `_audio_to_frames` (dup. preprocessors.py:60-78): ALWAYS pads
    `hop - N % hop` zeros (a whole hop when N % hop == 0); times = k / fps (f64)."""
    audio = np.asarray(audio)
    pad = hop - len(audio) % hop
    audio = np.pad(audio, [0, pad], mode="constant")
    n = len(audio) // hop
    return audio.reshape(n, hop), np.arange(n) / frames_per_second


def split_segments(frames: np.ndarray, times: np.ndarray, inputs_length: int):
    """t5 `split_tokens_to_inputs_length` [third-party, from memory]: consecutive
    chunks of `inputs_length` frames, last one keeps its true length."""
    out = []
    for s in range(0, len(frames), inputs_length):
        out.append((frames[s:s + inputs_length], times[s:s + inputs_length]))
    return out
