"""NoteSequence -> Standard MIDI File (the notebook ends with
`note_seq.sequence_proto_to_midi_file(est_ns, path)`).

note_seq / pretty_midi are not available here, so this is a from-scratch SMF type-1 writer (and a
reader, for round-trip tests) following their conventions: resolution = `ticks_per_quarter` (220),
one tempo of 120 qpm at tick 0, tick = round(time * resolution * qpm / 60), one track per
(instrument, program, is_drum), drums on channel 9.  PARITY UNPINNED against note_seq's writer.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

from .note_sequences import Note, NoteSequence

DEFAULT_QPM = 120.0


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track(events: List[Tuple[int, int, bytes]]) -> bytes:
    """events: (tick, order, payload); delta-encoded, End-of-Track appended."""
    events.sort(key=lambda e: (e[0], e[1]))
    data, last = bytearray(), 0
    for tick, _, payload in events:
        data += _vlq(tick - last) + payload
        last = tick
    data += b"\x00\xff\x2f\x00"
    return b"MTrk" + struct.pack(">I", len(data)) + bytes(data)


def note_sequence_to_midi_bytes(ns: NoteSequence, qpm: float = DEFAULT_QPM) -> bytes:
    res = ns.ticks_per_quarter or 220
    to_tick = lambda t: int(round(t * res * qpm / 60.0))  # noqa: E731
    tempo = int(round(60_000_000 / qpm))
    tracks = [_track([(0, 0, b"\xff\x51\x03" + struct.pack(">I", tempo)[1:])])]
    groups: Dict[Tuple[int, int, bool], List[Note]] = {}
    for n in ns.notes:
        groups.setdefault((n.instrument, n.program, bool(n.is_drum)), []).append(n)
    free = [c for c in range(16) if c != 9]
    for i, ((inst, program, drum), notes) in enumerate(sorted(groups.items())):
        ch = 9 if drum else free[i % len(free)]
        ev = [(0, 0, bytes([0xC0 | ch, program & 0x7F]))]
        for n in notes:
            on, off = to_tick(n.start_time), max(to_tick(n.end_time), to_tick(n.start_time) + 1)
            ev.append((on, 2, bytes([0x90 | ch, n.pitch & 0x7F, max(1, n.velocity) & 0x7F])))
            ev.append((off, 1, bytes([0x80 | ch, n.pitch & 0x7F, 0])))          # offs before ons at a tick
        tracks.append(_track(ev))
    header = b"MThd" + struct.pack(">IHHH", 6, 1, len(tracks), res)
    return header + b"".join(tracks)


def note_sequence_to_midi_file(ns: NoteSequence, path: str, qpm: float = DEFAULT_QPM) -> None:
    with open(path, "wb") as f:
        f.write(note_sequence_to_midi_bytes(ns, qpm))


sequence_proto_to_midi_file = note_sequence_to_midi_file       # note_seq's name for it


def midi_bytes_to_note_sequence(data: bytes) -> NoteSequence:
    """Minimal SMF reader (note on/off, program change, one tempo) -- enough to round-trip the writer."""
    assert data[:4] == b"MThd"
    _, fmt, ntrk, res = struct.unpack(">IHHH", data[4:14])
    pos, tempo = 14, 500000
    raw = []
    for _ in range(ntrk):
        assert data[pos:pos + 4] == b"MTrk"
        ln = struct.unpack(">I", data[pos + 4:pos + 8])[0]
        body, pos = data[pos + 8:pos + 8 + ln], pos + 8 + ln
        i, tick, status, program, open_notes = 0, 0, 0, {}, {}
        while i < len(body):
            d = 0
            while True:
                b = body[i]
                i += 1
                d = (d << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            tick += d
            if body[i] & 0x80:
                status = body[i]
                i += 1
            if status == 0xFF:
                kind, ln2 = body[i], body[i + 1]
                if kind == 0x51:
                    tempo = int.from_bytes(body[i + 2:i + 5], "big")
                i += 2 + ln2
                continue
            hi, ch = status & 0xF0, status & 0x0F
            if hi == 0xC0:
                program[ch] = body[i]
                i += 1
            elif hi in (0x80, 0x90):
                pitch, vel = body[i], body[i + 1]
                i += 2
                if hi == 0x90 and vel > 0:
                    open_notes.setdefault((ch, pitch), []).append((tick, vel))
                elif open_notes.get((ch, pitch)):
                    t0, v0 = open_notes[(ch, pitch)].pop(0)
                    raw.append((t0, tick, pitch, v0, program.get(ch, 0), ch == 9))
            else:
                i += 2 if hi not in (0xD0,) else 1
    sec = lambda t: t * tempo / 1e6 / res  # noqa: E731
    ns = NoteSequence(ticks_per_quarter=res)
    for t0, t1, pitch, vel, prog, drum in sorted(raw):
        ns.notes.append(Note(sec(t0), sec(t1), pitch, vel, prog, drum, 9 if drum else 0))
        ns.total_time = max(ns.total_time, sec(t1))
    return ns
