"""N3/N4 helpers around the path: MIDI export round trip, WAV ingest/resample, note matching."""
import itertools

import numpy as np

from mt3_amd import audio_io, metrics, midi_io
from mt3_amd.note_sequences import Note, NoteSequence


def _ns(rng, n=20):
    ns = NoteSequence()
    for _ in range(n):
        st = round(float(rng.uniform(0, 10)), 3)
        drum = rng.random() < 0.2
        ns.notes.append(Note(st, st + round(float(rng.uniform(0.02, 1.0)), 3), int(rng.integers(21, 108)),
                             int(rng.integers(1, 128)), 0 if drum else int(rng.choice([0, 24, 40])), drum,
                             9 if drum else 0))
        ns.total_time = max(ns.total_time, ns.notes[-1].end_time)
    return ns


def test_midi_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    ns = _ns(rng)
    p = tmp_path / "a.mid"
    midi_io.note_sequence_to_midi_file(ns, str(p))
    data = p.read_bytes()
    assert data[:4] == b"MThd" and data[12:14] == (220).to_bytes(2, "big")
    back = midi_io.midi_bytes_to_note_sequence(data)
    a = sorted((n.pitch, n.program, n.is_drum, n.velocity, n.start_time, n.end_time) for n in ns.notes)
    b = sorted((n.pitch, n.program, n.is_drum, n.velocity, n.start_time, n.end_time) for n in back.notes)
    assert len(a) == len(b)
    tick = 60.0 / (220 * 120.0)
    for x, y in zip(a, b):
        assert x[:4] == y[:4] and abs(x[4] - y[4]) <= tick and abs(x[5] - y[5]) <= 2 * tick


def test_wav_ingest_and_resample():
    t = np.arange(44100) / 44100.0
    x = 0.5 * np.sin(2 * np.pi * 440 * t)
    stereo = np.stack([x, x], 1)
    from scipy.io import wavfile
    import io
    buf = io.BytesIO()
    wavfile.write(buf, 44100, (stereo * 32767).astype(np.int16))
    y = audio_io.wav_data_to_samples(buf.getvalue())
    assert y.dtype == np.float32 and abs(len(y) - 16000) <= 1
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(len(y)) / 16000.0)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3
    z = audio_io.wav_data_to_samples(audio_io.samples_to_wav_data(y))
    assert np.abs(z - y).max() < 1e-4                      # 16 kHz in -> untouched apart from int16 quantisation


def test_note_matching_vs_brute_force():
    rng = np.random.default_rng(1)
    for _ in range(20):
        n = int(rng.integers(1, 6))
        ref_iv = np.sort(rng.uniform(0, 1, (n, 2)), 1)
        est_iv = ref_iv + rng.normal(0, 0.04, (n, 2))
        pitch = rng.integers(60, 62, n)
        est_pitch = pitch.copy()
        got = metrics.match_notes(ref_iv, pitch, est_iv, est_pitch)
        tol = np.maximum(0.05, 0.2 * (ref_iv[:, 1] - ref_iv[:, 0]))
        ok = (np.abs(ref_iv[:, None, 0] - est_iv[None, :, 0]) <= 0.05) & (pitch[:, None] == est_pitch[None]) & \
             (np.abs(ref_iv[:, None, 1] - est_iv[None, :, 1]) <= tol[:, None])
        best = max(sum(ok[i, p[i]] for i in range(n)) for p in itertools.permutations(range(n)))
        assert got == best
    a = NoteSequence(notes=[Note(0.0, 1.0, 60, 100), Note(1.0, 2.0, 62, 100), Note(0.0, 0.5, 36, 100, 0, True, 9)])
    b = NoteSequence(notes=[Note(0.03, 1.3, 60, 100), Note(1.2, 2.0, 62, 100)])
    s = metrics.transcription_scores(a, b)
    assert s["Onset F1"] == 0.5 and s["Onset + offset F1"] == 0.0 and s["Onset recall"] == 0.5


def test_midi_bytes_of_a_one_note_sequence_follow_the_smf_specification():
    """Pinned on the Standard MIDI File format itself, byte for byte (independent of this module's reader): header
    MThd / length 6 / format 1 / 2 tracks / division 220; track 0 = tempo 500,000 us per quarter (120 qpm) at tick 0;
    track 1 = program change, note-on at tick 220 (0.5 s x 220 x 120 / 60) as VLQ 0x81 0x5C, note-off 220 ticks
    later; every track ends with FF 2F 00."""
    ns = NoteSequence(notes=[Note(0.5, 1.0, 60, 100, 0, False, 0)], total_time=1.0)
    want = (b"MThd" + bytes([0, 0, 0, 6, 0, 1, 0, 2, 0, 220]) +
            b"MTrk" + bytes([0, 0, 0, 11]) + bytes([0x00, 0xFF, 0x51, 0x03, 0x07, 0xA1, 0x20, 0x00, 0xFF, 0x2F, 0x00]) +
            b"MTrk" + bytes([0, 0, 0, 17]) +
            bytes([0x00, 0xC0, 0x00, 0x81, 0x5C, 0x90, 0x3C, 0x64, 0x81, 0x5C, 0x80, 0x3C, 0x00, 0x00, 0xFF, 0x2F, 0x00]))
    assert midi_io.note_sequence_to_midi_bytes(ns) == want
    # a drum note goes to channel 9 and a note shorter than a tick still gets one tick
    d = NoteSequence(notes=[Note(0.0, 0.0005, 38, 127, 0, True, 9)], total_time=0.01)
    b = midi_io.note_sequence_to_midi_bytes(d)
    assert bytes([0x00, 0x99, 0x26, 0x7F, 0x01, 0x89, 0x26, 0x00]) in b
