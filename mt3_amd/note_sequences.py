"""Note containers and encoding specs (mirror of the decode half of mt3/note_sequences.py).

`NoteSequence` stands in for the `note_seq.NoteSequence` proto (fields the path
uses: notes[*].{start_time,end_time,pitch,velocity,program,is_drum,instrument},
total_time, ticks_per_quarter=220 -- note_sequences.py:281,307-310).  The note
state machine itself (note_sequences.py:262-408) lives in libmt3hip.so
(csrc/symbolic.cpp) and is selected by the spec objects below.
"""
from __future__ import annotations

import dataclasses
from typing import List

from . import _lib

DEFAULT_VELOCITY = 100
DEFAULT_NOTE_DURATION = 0.01
MIN_NOTE_DURATION = 0.01


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
class NoteSequence:
    notes: List[Note] = dataclasses.field(default_factory=list)
    total_time: float = 0.0
    ticks_per_quarter: int = 220


@dataclasses.dataclass(frozen=True)
class NoteEncodingSpecType:
    """Which decoder the C library runs (note_sequences.py:411-446)."""
    name: str
    spec_id: int


NoteOnsetEncodingSpec = NoteEncodingSpecType("NoteOnsetEncodingSpec", _lib.SPEC_ONSETS)
NoteEncodingSpec = NoteEncodingSpecType("NoteEncodingSpec", _lib.SPEC_NOTES)
NoteEncodingWithTiesSpec = NoteEncodingSpecType("NoteEncodingWithTiesSpec", _lib.SPEC_TIES)


# ---------------------------------------------------------------------------------------------
# Encode side (mirror of note_sequences.py:141-259): NoteSequence -> timed
# event data -> codec events.  Host integer work; used to synthesise valid token streams.
import itertools  # noqa: E402
from typing import Optional, Sequence, Tuple  # noqa: E402

from . import event_codec  # noqa: E402
from . import vocabularies  # noqa: E402


@dataclasses.dataclass
class NoteEventData:
    pitch: int
    velocity: Optional[int] = None
    program: Optional[int] = None
    is_drum: Optional[bool] = None
    instrument: Optional[int] = None


@dataclasses.dataclass
class NoteEncodingState:
    """velocity bin of every (pitch, program) seen so far (0 = off)."""
    active_pitches: dict = dataclasses.field(default_factory=dict)


def note_sequence_to_onsets(ns: NoteSequence):
    notes = sorted(ns.notes, key=lambda n: n.pitch)          # pitch = tie-break of the later stable sort
    return [n.start_time for n in notes], [NoteEventData(pitch=n.pitch) for n in notes]


def note_sequence_to_onsets_and_offsets(ns: NoteSequence):
    """offsets listed before onsets so that, at equal times, offsets sort first."""
    notes = sorted(ns.notes, key=lambda n: n.pitch)
    times = [n.end_time for n in notes] + [n.start_time for n in notes]
    values = ([NoteEventData(pitch=n.pitch, velocity=0) for n in notes] +
              [NoteEventData(pitch=n.pitch, velocity=n.velocity) for n in notes])
    return times, values


def note_sequence_to_onsets_and_offsets_and_programs(ns: NoteSequence):
    notes = sorted(ns.notes, key=lambda n: (n.is_drum, n.program, n.pitch))
    pitched = [n for n in notes if not n.is_drum]            # drums have no offsets
    times = [n.end_time for n in pitched] + [n.start_time for n in notes]
    values = ([NoteEventData(pitch=n.pitch, velocity=0, program=n.program, is_drum=False) for n in pitched] +
              [NoteEventData(pitch=n.pitch, velocity=n.velocity, program=n.program, is_drum=n.is_drum)
               for n in notes])
    return times, values


def note_event_data_to_events(state: Optional[NoteEncodingState], value: NoteEventData,
                              codec: event_codec.Codec) -> Sequence[event_codec.Event]:
    E = event_codec.Event
    if value.velocity is None:
        return [E("pitch", value.pitch)]
    vbin = vocabularies.velocity_to_bin(value.velocity, vocabularies.num_velocity_bins_from_codec(codec))
    if value.program is None:
        if state is not None:
            state.active_pitches[(value.pitch, 0)] = vbin
        return [E("velocity", vbin), E("pitch", value.pitch)]
    if value.is_drum:
        return [E("velocity", vbin), E("drum", value.pitch)]
    if state is not None:
        state.active_pitches[(value.pitch, int(value.program))] = vbin
    return [E("program", value.program), E("velocity", vbin), E("pitch", value.pitch)]


def note_encoding_state_to_events(state: NoteEncodingState) -> Sequence[event_codec.Event]:
    """program/pitch pairs of the sounding notes (sorted by program, then pitch) + the `tie` marker."""
    E = event_codec.Event
    out = []
    for pitch, program in sorted(state.active_pitches, key=lambda k: (k[1], k[0])):
        if state.active_pitches[(pitch, program)]:
            out += [E("program", program), E("pitch", pitch)]
    out.append(E("tie", 0))
    return out


# what the three specs use on the encode side (note_sequences.py:416-446)
ENCODING_FNS = {
    "NoteOnsetEncodingSpec": (lambda: None, note_event_data_to_events, None),
    "NoteEncodingSpec": (lambda: None, note_event_data_to_events, None),
    "NoteEncodingWithTiesSpec": (NoteEncodingState, note_event_data_to_events, note_encoding_state_to_events),
}
