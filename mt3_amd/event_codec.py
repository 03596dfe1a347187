"""Event codec on top of the C library (mirror of mt3/event_codec.py:21-112).

`Codec` keeps the reference's constructor and methods; the index arithmetic is
done by libmt3hip.so (`mt3_codec_encode_event` / `mt3_codec_decode_event`), the
same table the note decoder uses, so both can never disagree.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import List, Tuple

from . import _lib


@dataclasses.dataclass
class EventRange:
    type: str
    min_value: int
    max_value: int


@dataclasses.dataclass
class Event:
    type: str
    value: int


def _type_id(name: str) -> int:
    try:
        return _lib.EVENT_TYPE_NAMES.index(name)
    except ValueError:
        raise ValueError(f"Unknown event type: {name}") from None


class Codec:
    """shift range first (ids 0..max_shift_steps), then `event_ranges` in order."""

    def __init__(self, max_shift_steps: int, steps_per_second: float, event_ranges: List[EventRange]):
        self.steps_per_second = steps_per_second
        self._shift_range = EventRange("shift", 0, max_shift_steps)
        self._event_ranges = [self._shift_range] + list(event_ranges)
        names = [r.type for r in self._event_ranges]
        assert len(names) == len(set(names)), "event types must be unique"
        if len(self._event_ranges) > 8:
            raise ValueError("at most 8 event ranges")
        self.desc = _lib.CodecDesc()
        self.desc.steps_per_second = float(steps_per_second)
        self.desc.num_ranges = len(self._event_ranges)
        for i, r in enumerate(self._event_ranges):
            self.desc.ranges[i] = _lib.EventRange(_type_id(r.type), int(r.min_value), int(r.max_value))
        self._lib = _lib.load()

    @classmethod
    def from_desc(cls, desc: "_lib.CodecDesc") -> "Codec":
        rs = [EventRange(_lib.EVENT_TYPE_NAMES[desc.ranges[i].type], desc.ranges[i].min_value,
                         desc.ranges[i].max_value) for i in range(1, desc.num_ranges)]
        return cls(desc.ranges[0].max_value, desc.steps_per_second, rs)

    @property
    def num_classes(self) -> int:
        n = self._lib.mt3_codec_num_classes(C.byref(self.desc))
        if n < 0:
            _lib.check(n)
        return n

    @property
    def max_shift_steps(self) -> int:
        return self._shift_range.max_value

    def is_shift_event_index(self, index: int) -> bool:
        return self._shift_range.min_value <= index <= self._shift_range.max_value

    def event_type_range(self, event_type: str) -> Tuple[int, int]:
        off = 0
        for r in self._event_ranges:
            if r.type == event_type:
                return off, off + (r.max_value - r.min_value)
            off += r.max_value - r.min_value + 1
        raise ValueError(f"Unknown event type: {event_type}")

    def encode_event(self, event: Event) -> int:
        out = C.c_int32()
        rc = self._lib.mt3_codec_encode_event(C.byref(self.desc), _type_id(event.type), int(event.value), C.byref(out))
        if rc != 0:
            raise ValueError(f"cannot encode {event}: {self._lib.mt3_last_error().decode()}")
        return out.value

    def decode_event_index(self, index: int) -> Event:
        t, v = C.c_int32(), C.c_int32()
        rc = self._lib.mt3_codec_decode_event(C.byref(self.desc), int(index), C.byref(t), C.byref(v))
        if rc != 0:
            raise ValueError(f"Unknown event index: {index}")
        return Event(_lib.EVENT_TYPE_NAMES[t.value], v.value)
