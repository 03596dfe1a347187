#!/usr/bin/env python3
"""Generate tests/golden/symbolic_golden.json by running the REAL reference modules.

Runs only in the build container (needs /root/reference).  The reference's
symbolic stage is pure Python but imports TensorFlow / seqio / t5 / note_seq /
absl / pretty_midi at module scope; none is installed here, and none is used by
the functions on the decode path.  We therefore register minimal stand-in
modules (constants, a list-backed NoteSequence, an identity `map_over_dataset`,
a bare `Vocabulary` base class) and import the reference's own

    mt3/event_codec.py  mt3/vocabularies.py  mt3/run_length_encoding.py
    mt3/note_sequences.py  mt3/metrics_utils.py

unmodified from /root/reference.  Every number in the fixture is produced by
the reference's code: `Codec.encode_event/decode_event_index`,
`GenericTokenVocabulary._decode/_encode`, `velocity_to_bin/bin_to_velocity`,
`encode_and_index_events` (encode side, used to synthesise *valid* token
streams from random notes), `run_length_encoding.decode_events` and
`metrics_utils.event_predictions_to_ns` for the three encoding specs.

Usage:  python tests/golden/make_symbolic_golden.py   (rewrites the .json)
"""
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True   # never write into /root/reference
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "symbolic_golden.json")


# ------------------------------------------------------------------ stand-in modules
class _Note:
    def __init__(self, **kw):
        self.start_time = 0.0
        self.end_time = 0.0
        self.pitch = 0
        self.velocity = 0
        self.program = 0
        self.is_drum = False
        self.instrument = 0
        for k, v in kw.items():
            setattr(self, k, v)


class _Notes(list):
    def add(self, **kw):
        n = _Note(**kw)
        self.append(n)
        return n


class _NoteSequence:
    def __init__(self, ticks_per_quarter=0):
        self.ticks_per_quarter = ticks_per_quarter
        self.total_time = 0.0
        self.notes = _Notes()


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Logging:
        @staticmethod
        def info(*a, **k):
            pass
        warning = info

    absl = mod("absl")
    absl.logging = mod("absl.logging", info=_Logging.info, warning=_Logging.info)

    mod("note_seq", NoteSequence=_NoteSequence, MIN_MIDI_PITCH=0, MAX_MIDI_PITCH=127,
        MIN_MIDI_PROGRAM=0, MAX_MIDI_PROGRAM=127, MAX_MIDI_VELOCITY=127)

    class _Vocabulary:                       # seqio.Vocabulary: only what the reference calls
        def __init__(self, extra_ids=0):
            self._extra_ids = extra_ids

        @property
        def extra_ids(self):
            return self._extra_ids

        @property
        def vocab_size(self):
            return self._base_vocab_size + self._extra_ids

    mod("seqio", Vocabulary=_Vocabulary, map_over_dataset=lambda f: f)
    t5 = mod("t5")
    t5.data = mod("t5.data", DEFAULT_EXTRA_IDS=100)
    tf = mod("tensorflow", Tensor=type("Tensor", (), {}))
    tf.data = types.SimpleNamespace(Dataset=object)
    mod("pretty_midi")
    mod("sklearn")
    # `mt3/__init__.py` imports the whole package (datasets, tasks, ... -> real TF);
    # register an empty package object whose __path__ points at the reference so
    # that `from mt3 import X` loads mt3/X.py itself without running __init__.py.
    pkg = mod("mt3")
    pkg.__path__ = [os.path.join(REF, "mt3")]


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    _install_stubs()
    from mt3 import event_codec, vocabularies, run_length_encoding, note_sequences, metrics_utils

    rng = np.random.default_rng(20260923)
    out = {"generator": "tests/golden/make_symbolic_golden.py",
           "reference": "magenta/mt3 @ /root/reference (real modules, stubbed third-party imports)"}

    # ---- codec tables for both presets + the 100-step test codec
    codecs = {}
    for name, bins in (("mt3", 1), ("ismir2021", 127)):
        c = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=bins))
        v = vocabularies.vocabulary_from_codec(c)
        probe = sorted(set([0, 1, 500, 1000, 1001, 1128, 1129, 1130, 1131, 1132, 1259, 1260,
                            c.num_classes - 1] + [int(x) for x in rng.integers(0, c.num_classes, 40)]))
        probe = [p for p in probe if p < c.num_classes]
        codecs[name] = {
            "num_velocity_bins": bins,
            "num_classes": c.num_classes,
            "vocab_size": v.vocab_size,
            "num_embeddings": vocabularies.num_embeddings(v),
            "type_ranges": {t: list(c.event_type_range(t))
                            for t in ("shift", "pitch", "velocity", "tie", "program", "drum")},
            "decode_probe": [[p, c.decode_event_index(p).type, c.decode_event_index(p).value]
                             for p in probe],
            "vocab_decode_in": [0, 1, 2, 3, 4, c.num_classes + 2, c.num_classes + 3, v.vocab_size - 1,
                                v.vocab_size, 9, 1, 7],
        }
        codecs[name]["vocab_decode_out"] = v._decode(codecs[name]["vocab_decode_in"])
        codecs[name]["velocity_roundtrip"] = [
            [vel, vocabularies.velocity_to_bin(vel, bins),
             vocabularies.bin_to_velocity(vocabularies.velocity_to_bin(vel, bins), bins)]
            for vel in (0, 1, 2, 63, 64, 100, 126, 127)]
    out["codecs"] = codecs

    # ---- decode cases: random notes -> reference ENCODER -> per-segment tokens (+ corruption)
    specs = {"onsets": note_sequences.NoteOnsetEncodingSpec,
             "notes": note_sequences.NoteEncodingSpec,
             "ties": note_sequences.NoteEncodingWithTiesSpec}
    cases = []
    for case_id in range(36):
        mode = ("ties", "notes", "onsets")[case_id % 3]
        bins = 1 if mode == "ties" else (127 if case_id % 2 else 1)
        codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=bins))
        spec = specs[mode]
        n_seg = int(rng.integers(1, 6))
        seg_len = 2.048
        total = n_seg * seg_len
        # random notes
        ns = _NoteSequence(ticks_per_quarter=220)
        for _ in range(int(rng.integers(3, 40))):
            st = float(rng.uniform(0, total - 0.05))
            en = float(min(total, st + rng.uniform(0.02, 3.0)))
            drum = bool(mode == "ties" and rng.random() < 0.15)
            ns.notes.add(start_time=st, end_time=en, pitch=int(rng.integers(21, 109)),
                         velocity=int(rng.integers(1, 128)),
                         program=0 if mode != "ties" or drum else int(rng.choice([0, 24, 40, 41, 73])),
                         is_drum=drum)
        if mode == "onsets":
            times, values = note_sequences.note_sequence_to_onsets(ns)
        elif mode == "notes":
            times, values = note_sequences.note_sequence_to_onsets_and_offsets(ns)
        else:
            times, values = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(ns)
        frame_times = np.arange(int(total * 125)) / 125.0
        (events, ev_start, ev_end, state_events, state_idx) = run_length_encoding.encode_and_index_events(
            state=spec.init_encoding_state_fn(), event_times=times, event_values=values,
            encode_event_fn=spec.encode_event_fn, codec=codec, frame_times=frame_times,
            encoding_state_to_events_fn=spec.encoding_state_to_events_fn)
        preds = []
        for s in range(n_seg):
            f0, f1 = s * 256, min((s + 1) * 256, len(frame_times)) - 1
            seg = [int(e) for e in events[ev_start[f0]:ev_end[f1]]]
            # run-length encode the unit shifts the way run_length_encode_shifts_fn does
            # (absolute steps since segment start, emitted before each non-shift event)
            toks, steps, total_steps = [], 0, 0
            for e in seg:
                if codec.is_shift_event_index(e):
                    steps += 1
                    total_steps += 1
                else:
                    if steps > 0:
                        rem = total_steps
                        while rem > 0:
                            o = min(codec.max_shift_steps, rem)
                            toks.append(o)
                            rem -= o
                        steps = 0
                    toks.append(e)
            if mode == "ties":
                st_ev = [int(e) for e in state_events[state_idx[f0]:]]
                tie_id = codec.encode_event(event_codec.Event("tie", 0))
                st_ev = st_ev[:st_ev.index(tie_id) + 1] if tie_id in st_ev else [tie_id]
                toks = st_ev + toks
            # corruption: some cases get junk tokens / out-of-range / shuffles
            if case_id % 4 == 3 and toks:
                for _ in range(int(rng.integers(1, 6))):
                    pos = int(rng.integers(0, len(toks) + 1))
                    toks.insert(pos, int(rng.choice([-2, -1 - 1, codec.num_classes, codec.num_classes + 50,
                                                     int(rng.integers(0, codec.num_classes))])))
            start_time = frame_times[f0]
            start_time -= start_time % (1 / codec.steps_per_second)
            preds.append({"est_tokens": np.array(toks, np.int32), "start_time": float(start_time),
                          "raw_inputs": np.zeros((0,), np.float32)})
        if case_id % 5 == 4:           # out-of-order segment list: the combiner must sort
            perm = rng.permutation(len(preds))
            preds = [preds[i] for i in perm]
        res = metrics_utils.event_predictions_to_ns(preds, codec=codec, encoding_spec=spec)
        ens = res["est_ns"]
        cases.append({
            "mode": mode, "num_velocity_bins": bins,
            "segments": [{"start_time": p["start_time"], "tokens": [int(t) for t in p["est_tokens"]]}
                         for p in preds],
            "invalid": int(res["est_invalid_events"]), "dropped": int(res["est_dropped_events"]),
            "total_time": float(ens.total_time),
            "notes": [[float(n.start_time), float(n.end_time), int(n.pitch), int(n.velocity),
                       int(n.program), bool(n.is_drum), int(n.instrument)] for n in ens.notes],
        })
    # ---- fully random token soup (exercises every error branch)
    for case_id in range(24):
        mode = ("ties", "notes", "onsets")[case_id % 3]
        bins = 1 if case_id % 2 == 0 else 127
        codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=bins))
        spec = specs[mode]
        n_seg = int(rng.integers(1, 5))
        preds = []
        for s in range(n_seg):
            n = int(rng.integers(0, 120))
            kinds = rng.random(n)
            toks = np.where(kinds < 0.35, rng.integers(1, 300, n),
                            rng.integers(-3, codec.num_classes + 20, n)).astype(np.int32)
            preds.append({"est_tokens": toks, "start_time": float(s * 2.048 - (s * 2.048) % 0.01),
                          "raw_inputs": np.zeros((0,), np.float32)})
        res = metrics_utils.event_predictions_to_ns(preds, codec=codec, encoding_spec=spec)
        ens = res["est_ns"]
        cases.append({
            "mode": mode, "num_velocity_bins": bins,
            "segments": [{"start_time": p["start_time"], "tokens": [int(t) for t in p["est_tokens"]]}
                         for p in preds],
            "invalid": int(res["est_invalid_events"]), "dropped": int(res["est_dropped_events"]),
            "total_time": float(ens.total_time),
            "notes": [[float(n.start_time), float(n.end_time), int(n.pitch), int(n.velocity),
                       int(n.program), bool(n.is_drum), int(n.instrument)] for n in ens.notes],
        })
    out["decode_cases"] = cases

    # ---- encode side (N2): the reference's encode_and_index_events on random note sets
    enc_cases = []
    for case_id in range(12):
        mode = ("ties", "notes", "onsets")[case_id % 3]
        bins = 1 if mode == "ties" else 127
        codec = vocabularies.build_codec(vocabularies.VocabularyConfig(num_velocity_bins=bins))
        spec = specs[mode]
        total = float(rng.uniform(1.0, 5.0))
        notes = []
        for _ in range(int(rng.integers(1, 25))):
            st = round(float(rng.uniform(0, total - 0.05)), int(rng.integers(2, 5)))
            en = round(float(min(total, st + rng.uniform(0.0, 2.0))), int(rng.integers(2, 5)))
            drum = bool(mode == "ties" and rng.random() < 0.2)
            notes.append(dict(start_time=st, end_time=max(en, st), pitch=int(rng.integers(21, 109)),
                              velocity=int(rng.integers(1, 128)),
                              program=0 if mode != "ties" or drum else int(rng.choice([0, 24, 40])), is_drum=drum))
        ns = _NoteSequence(ticks_per_quarter=220)
        for n in notes:
            ns.notes.add(**n)
        if mode == "onsets":
            times, values = note_sequences.note_sequence_to_onsets(ns)
        elif mode == "notes":
            times, values = note_sequences.note_sequence_to_onsets_and_offsets(ns)
        else:
            times, values = note_sequences.note_sequence_to_onsets_and_offsets_and_programs(ns)
        frame_times = np.arange(int(total * 125) + 1) / 125.0
        ev, es, ee, se, si = run_length_encoding.encode_and_index_events(
            state=spec.init_encoding_state_fn(), event_times=times, event_values=values,
            encode_event_fn=spec.encode_event_fn, codec=codec, frame_times=frame_times,
            encoding_state_to_events_fn=spec.encoding_state_to_events_fn)
        enc_cases.append({"mode": mode, "num_velocity_bins": bins, "notes": notes, "n_frames": len(frame_times),
                          "events": [int(x) for x in ev], "start": [int(x) for x in es], "end": [int(x) for x in ee],
                          "state_events": [int(x) for x in se], "state_idx": [int(x) for x in si]})
    out["encode_cases"] = enc_cases
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "cases;",
          sum(len(c["notes"]) for c in cases), "notes")


if __name__ == "__main__":
    main()
