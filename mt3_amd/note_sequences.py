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
