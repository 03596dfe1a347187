"""Drop-in inference surface of MT3 on the MI355X engine.

`InferenceModel` mirrors the class the reference defines in its Colab notebook
(mt3/colab/music_transcription_with_transformers.ipynb, cell "Imports and
Definitions"): same constructor arguments, attributes (`inputs_length`,
`outputs_length`, `batch_size`, `sequence_length`, `encoding_spec`,
`spectrogram_config`, `codec`, `vocabulary`, `input_shapes`) and methods
(`restore_from_checkpoint`, `predict_tokens`, `__call__`, `audio_to_dataset`,
`preprocess`, `postprocess`, `_trim_eos`).  `write_inferences_to_file` mirrors
mt3/inference.py:34-138 (the t5x `infer` write_fn).

Differences that are deliberate: the t5x/gin/tf.data plumbing is gone -- segments
are cut and padded on the host exactly as the reference's preprocessors do
(`_audio_to_frames`, split into `inputs_length`-frame chunks, zero rows after the
log for a short last segment), everything numeric runs in libmt3hip.so.  `batch_size`
stays the reference's 8 as an ATTRIBUTE (`input_shapes`, NB:190), but the engine behind
`predict_tokens` is sized to the JOB -- up to `max_slots` decode slots, grown lazily -- and
runs mt3_engine_transcribe: a 10-minute file is ONE engine call whose finished rows are
refilled with the file's next segments, not 37 batch-synchronous calls of 8 rows
(`schedule="batch"` keeps the reference's loop for comparison); the log-mel stays on
the device between `preprocess` and `predict_tokens`.  Decoding
defaults to `decoding="beam1"`: the selection rule of t5x beam search with one beam and
alpha 0.6, which is what the reference's predict_batch_with_aux runs;
`decoding="greedy"` stops a row at its first arg-max EOS.
"""
from __future__ import annotations

import json
import os
import re
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import metrics_utils
from . import network
from . import note_sequences
from . import spectrograms
from . import vocabularies

SAMPLE_RATE = 16000


def trim_eos(tokens: Sequence[int]) -> np.ndarray:
    """tasks.trim_eos (mt3/tasks.py:58-63) == InferenceModel._trim_eos."""
    tokens = np.array(tokens, np.int32)
    if vocabularies.DECODED_EOS_ID in tokens:
        tokens = tokens[: np.argmax(tokens == vocabularies.DECODED_EOS_ID)]
    return tokens


class InferenceModel(object):
    """Wrapper of the MI355X engine for music transcription."""

    def __init__(self, checkpoint_path, model_type="mt3", *, config: Optional[network.T5Config] = None,
                 dtype: str = "float32", batch_size: int = 8, early_exit: bool = True,
                 decoding: str = "beam1", max_slots: int = 256, schedule: str = "refill"):
        """dtype: 'float32' (default) = the reference's own precision (gin/model.gin:50 restores and runs
        float32): f32 MFMA operands, f32 K/V cache, token-exact against the oracle.  'bfloat16' is the explicit
        opt-in fast path (bf16 operands and caches, f32 accumulation / residual / softmax; what bench.py times;
        logits within 3e-2 rel-L2 of f32 at every cache depth, tests/test_gpu_parity_deep.py).
        max_slots: the most decode slots the engine behind `predict_tokens` may hold (its K/V caches are allocated for
        that many rows: 3.2 GB per 64 slots in f32 at the MT3 shape); the engine starts at `batch_size` slots and is rebuilt
        with more (next power of two) when a job has more segments.  schedule: 'refill' = one mt3_engine_transcribe call
        per job (in-flight batching; needs early_exit); 'batch' = the reference's loop, one batch-synchronous engine call
        per `batch_size` segments (NB:295-301)."""
        if model_type == "ismir2021":
            num_velocity_bins = 127
            self.encoding_spec = note_sequences.NoteEncodingSpec
            self.inputs_length = 512
        elif model_type == "mt3":
            num_velocity_bins = 1
            self.encoding_spec = note_sequences.NoteEncodingWithTiesSpec
            self.inputs_length = 256
        else:
            raise ValueError("unknown model_type: %s" % model_type)

        self.batch_size = batch_size             # the reference's 8 (NB:190): what input_shapes reports
        if schedule not in ("refill", "batch"):
            raise ValueError("schedule must be 'refill' or 'batch', got %r" % (schedule,))
        self.schedule = schedule
        self.max_slots = max(int(max_slots), batch_size)
        self.outputs_length = 1024
        self.sequence_length = {"inputs": self.inputs_length, "targets": self.outputs_length}
        self.early_exit = early_exit
        if decoding not in ("beam1", "greedy"):
            raise ValueError("decoding must be 'beam1' or 'greedy', got %r" % (decoding,))
        self.decoding = decoding

        self.spectrogram_config = spectrograms.SpectrogramConfig()
        self.codec = vocabularies.build_codec(
            vocab_config=vocabularies.VocabularyConfig(num_velocity_bins=num_velocity_bins))
        self.vocabulary = vocabularies.vocabulary_from_codec(self.codec)
        self.output_features = {"inputs": None, "targets": self.vocabulary}

        base = config or network.T5Config()
        self.model_config = network.T5Config(**{
            **{f: getattr(base, f) for f in base.__dataclass_fields__},
            "vocab_size": vocabularies.num_embeddings(self.vocabulary), "dtype": dtype,
            "input_depth": spectrograms.input_depth(self.spectrogram_config)})
        self._params = None
        self.model = network.Transformer(self.model_config, input_length=self.inputs_length,
                                         max_decode_length=self.outputs_length, max_batch=self.batch_size)
        self.restore_from_checkpoint(checkpoint_path)

    @property
    def input_shapes(self):
        return {"encoder_input_tokens": (self.batch_size, self.inputs_length),
                "decoder_input_tokens": (self.batch_size, self.outputs_length)}

    def restore_from_checkpoint(self, checkpoint_path):
        """Weights: a t5x checkpoint DIRECTORY (what the reference restores: msgpack index + one zarr
        array per `target.*` parameter, read by mt3_amd.checkpoints), a flat `.npz` (names = the
        reference's Flax tree joined by '/'; or the int8 form of checkpoints.save_compact_npz), a dict of arrays, or 'random:<seed>' / None for the
        reference's initialisers (no checkpoint ships with the repo)."""
        rnd = None if checkpoint_path is None or isinstance(checkpoint_path, dict) else \
            re.fullmatch(r"random(?::(\d+))?", str(checkpoint_path))
        if isinstance(checkpoint_path, dict):
            params = checkpoint_path
        elif checkpoint_path is not None and os.path.isdir(str(checkpoint_path)):      # real paths win over 'random'
            from . import checkpoints
            params = checkpoints.load_t5x_checkpoint(str(checkpoint_path),
                                                     expected=network.param_shapes(self.model_config))
        elif checkpoint_path is not None and str(checkpoint_path).endswith(".npz") and \
                os.path.exists(str(checkpoint_path)):
            with np.load(str(checkpoint_path)) as z:
                compact = any(k.endswith("|q") for k in z.files)      # checkpoints.save_compact_npz: int8 + per-column scales
                params = None if compact else {k: z[k] for k in z.files}
            if compact:
                from . import checkpoints
                params = checkpoints.load_compact_npz(str(checkpoint_path))
        elif checkpoint_path is None or rnd:
            params = network.init_random_params(self.model_config, seed=int(rnd.group(1) or 0) if rnd else 0)
        else:
            raise ValueError("unsupported checkpoint %r: pass a t5x checkpoint directory, a flat .npz, a dict, "
                             "or 'random:<seed>'"
                             % (checkpoint_path,))
        self._params = params                    # kept: the engine is rebuilt with more slots when a job asks for them
        self.model.load_params(params)

    @property
    def engine_slots(self) -> int:
        return self.model.max_batch

    def _ensure_slots(self, n_segments: int):
        """the engine sized to the job: min(n_segments, max_slots) decode slots, in powers of two so that a run of files
        of similar length rebuilds it once (the attribute `batch_size` does not change)"""
        want = min(max(n_segments, self.batch_size), self.max_slots)
        if want <= self.model.max_batch:
            return
        slots = self.batch_size
        while slots < want:
            slots *= 2
        slots = min(slots, self.max_slots)
        # the old engine goes FIRST: two engines' K/V caches must never be resident together (12.8 GB each in f32 at 256
        # slots, MT3 shape) -- Transformer.__del__ destroys the engine as soon as the last reference is dropped
        self.model = None
        self.model = network.Transformer(self.model_config, input_length=self.inputs_length,
                                         max_decode_length=self.outputs_length, max_batch=slots)
        self.model.load_params(self._params)

    # ------------------------------------------------------------------ model call
    def predict_tokens(self, batch: Dict[str, Any], seed: int = 0) -> np.ndarray:
        """batch['encoder_input_tokens']: f32 [B, T, 512] (numpy or CUDA tensor) -> int32 [B, 1024]
        with -1 from EOS on and -2 for invalid ids (vocabulary.decode_tf)."""
        import torch
        x = batch["encoder_input_tokens"]
        x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x, np.float32))
        x = x.cuda()
        beam1 = self.decoding == "beam1"
        if self.schedule == "refill" and self.early_exit:
            # ONE engine call for the whole job: finished rows restart on the job's next segments
            self._ensure_slots(x.shape[0])
            self.rows_per_engine_call = [int(x.shape[0])]
            return self.vocabulary.decode_tf(self.model.transcribe(x, beam1=beam1)).cpu().numpy()
        out = []
        step = min(self.batch_size, self.model.max_batch)
        self.rows_per_engine_call = []
        for s in range(0, x.shape[0], step):
            self.model.encode(x[s:s + step])
            ids = self.model.decode(early_exit=self.early_exit, beam1=beam1)
            out.append(self.vocabulary.decode_tf(ids))
            self.rows_per_engine_call.append(int(min(step, x.shape[0] - s)))
        return torch.cat(out, 0).cpu().numpy()

    def __call__(self, audio):
        """1-d numpy array of 16 kHz samples -> NoteSequence."""
        ds = self.audio_to_dataset(audio)
        examples = self.preprocess(ds, host_inputs=False)
        # the frontend kernel has already written the feature converter's form of every segment -- [T, 512] rows, 0.0
        # after a short last segment's frames (mt3/models.py:48-98 via models.convert_features) -- and it is still on the
        # device: no host round trip between preprocess and predict_tokens
        batch, self._logmel_dev = {"encoder_input_tokens": self._logmel_dev}, None
        tokens = self.predict_tokens(batch)
        predictions = [self.postprocess(t, ex) for t, ex in zip(tokens, examples)]
        result = metrics_utils.event_predictions_to_ns(predictions, codec=self.codec,
                                                       encoding_spec=self.encoding_spec)
        return result["est_ns"]

    def transcribe_many(self, audios: Sequence[Any]) -> List[Any]:
        """Several files as ONE job (no counterpart in the notebook, which loops `model(audio)` over files): the segments of
        all files go through the engine's decode slots in one refilled call -- a finished slot restarts on the next
        segment, whichever file it belongs to -- and every file's tokens then become notes on their own (the note state
        machine is sequential within a file and independent across files, mt3/metrics_utils.py:92-116).  Returns one
        NoteSequence per file, each identical to `self(audio)`."""
        import torch
        per_file, feats = [], []
        for audio in audios:
            examples = self.preprocess(self.audio_to_dataset(audio), host_inputs=False)
            per_file.append(examples)
            feats.append(self._logmel_dev)
        self._logmel_dev = None
        if not per_file:
            return []
        tokens = self.predict_tokens({"encoder_input_tokens": torch.cat(feats, 0)})
        out, at = [], 0
        for examples in per_file:
            preds = [self.postprocess(t, ex) for t, ex in zip(tokens[at:at + len(examples)], examples)]
            at += len(examples)
            out.append(metrics_utils.event_predictions_to_ns(preds, codec=self.codec,
                                                             encoding_spec=self.encoding_spec)["est_ns"])
        return out

    # ------------------------------------------------------------------ host preprocessing
    def audio_to_dataset(self, audio):
        frames, frame_times = self._audio_to_frames(audio)
        return {"inputs": frames, "input_times": frame_times}

    def _audio_to_frames(self, audio):
        frame_size = self.spectrogram_config.hop_width
        audio = np.asarray(audio)
        padding = [0, frame_size - len(audio) % frame_size]      # always pads (a full hop if aligned)
        audio = np.pad(audio, padding, mode="constant")
        frames = spectrograms.split_audio(audio, self.spectrogram_config)
        num_frames = len(audio) // frame_size
        times = np.arange(num_frames) / self.spectrogram_config.frames_per_second
        return frames, times

    def preprocess(self, ds, host_inputs: bool = True) -> List[Dict[str, Any]]:
        """split_tokens_to_inputs_length + add_dummy_targets + compute_spectrograms
        (preprocessors.py:53-57,613-618), batched over all segments in one kernel launch.
        host_inputs=False (what `__call__` / `transcribe_many` pass): the log-mel is left on the device only -- the examples'
        'inputs' are None -- because the engine reads it there and nothing on that path looks at the host copy."""
        import torch
        frames, times = ds["inputs"], ds["input_times"]
        T, hop = self.inputs_length, self.spectrogram_config.hop_width
        n_seg = -(-len(frames) // T)
        audio = np.zeros((n_seg, T * hop), np.float32)
        counts = []
        for s in range(n_seg):
            chunk = frames[s * T:(s + 1) * T]
            audio[s, : chunk.size] = chunk.reshape(-1)
            counts.append(len(chunk))
        logmel_dev = spectrograms.compute_spectrogram_batch(torch.from_numpy(audio).cuda(), counts, self.spectrogram_config)
        # host_inputs=False: the device tensor is kept for the predict_tokens call that follows (and dropped by it);
        # otherwise the examples carry host arrays, as the reference's preprocess returns them, and nothing stays pinned
        self._logmel_dev = None if host_inputs else logmel_dev
        logmel = logmel_dev.cpu().numpy() if host_inputs else None
        return [{"inputs": logmel[s, : counts[s]] if host_inputs else None, "input_times": times[s * T:(s + 1) * T],
                 "raw_inputs": audio[s, : counts[s] * hop], "targets": np.zeros((0,), np.int32)}
                for s in range(n_seg)]

    def postprocess(self, tokens, example):
        tokens = self._trim_eos(tokens)
        start_time = example["input_times"][0]
        start_time -= start_time % (1 / self.codec.steps_per_second)   # float64, as in the notebook
        return {"est_tokens": tokens, "start_time": start_time, "raw_inputs": []}

    @staticmethod
    def _trim_eos(tokens):
        return trim_eos(tokens)


def _varint(buf: bytes, i: int):
    v, shift = 0, 0
    while True:
        if i >= len(buf) or shift > 63:
            raise ValueError("note_sequence_id: truncated or malformed protobuf varint")
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, i
        shift += 7


def note_sequence_id(sequence) -> str:
    """`NoteSequence.id` of the task example's 'sequence' feature (mt3/inference.py:83-86,133: the reference parses the
    serialized proto with note_seq and writes `ref_ns.id`).  Accepts an object with an `.id`, a str (taken as the id),
    or the SERIALIZED proto bytes: `string id = 1` of note_seq's music.proto [field number from memory], read straight
    off the protobuf wire format (tag = field << 3 | wire type; 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32)."""
    if hasattr(sequence, "id"):
        return str(sequence.id)
    if isinstance(sequence, str):
        return sequence
    buf = bytes(sequence)
    i = 0
    while i < len(buf):
        tag, i = _varint(buf, i)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            _, i = _varint(buf, i)
        elif wt == 1:
            i += 8
        elif wt == 5:
            i += 4
        elif wt == 2:
            n, i = _varint(buf, i)
            if i + n > len(buf):
                raise ValueError("note_sequence_id: length-delimited field runs past the end of the buffer")
            if field == 1:
                return buf[i:i + n].decode("utf-8")
            i += n
        else:
            raise ValueError("note_sequence_id: unsupported protobuf wire type %d" % wt)
    return ""                                     # proto3 default: an unset id is the empty string


class MissingSequenceError(AssertionError, ValueError):
    """A track of the task dataset has no NoteSequence (or a NoteSequence has no track).  The reference trips a bare
    `assert` there (mt3/inference.py:114), so callers that catch AssertionError keep working; it is a real exception
    here, not an assert statement, so `python -O` does not turn it into a KeyError further down."""


def write_inferences_to_file(path: str, inferences: Sequence[Any], task_ds, mode: str, vocabulary=None,
                             vocab_config=None, onsets_only=None, use_ties=None) -> None:
    """mt3/inference.py:34-138: one JSON line {"id", "est_notes": [...]} per track.
    `task_ds`: iterable of dicts with 'input_times', 'unique_id' (and optionally 'raw_inputs', 'sequence');
    `inferences`: one int32 id row per example (model ids, before decode_tf).
    "id": as in the reference (:83-86,133) the `id` of the NoteSequence the track's examples carry in 'sequence'
    (first non-empty one per `unique_id`; every track must then have one: the reference asserts it, :114); a task
    dataset WITHOUT a 'sequence' feature (plain inference, no ground truth) writes the `unique_id` itself."""
    if mode == "score":
        raise ValueError("`score` mode currently not supported in MT3")
    if not vocabulary:
        raise ValueError("`vocabulary` parameter required in `predict` mode")
    if vocab_config is None or onsets_only is None or use_ties is None:
        raise ValueError("vocab_config, onsets_only and use_ties are required")
    if onsets_only and use_ties:
        raise ValueError("ties not compatible with onset-only transcription")
    if onsets_only:
        encoding_spec = note_sequences.NoteOnsetEncodingSpec
    elif not use_ties:
        encoding_spec = note_sequences.NoteEncodingSpec
    else:
        encoding_spec = note_sequences.NoteEncodingWithTiesSpec
    codec = vocabularies.build_codec(vocab_config)

    def first(x):
        x = np.asarray(x)
        return x.reshape(-1)[0] if x.ndim else x[()]

    predictions, ref_ids, any_sequence = [], {}, False
    for inp, output in zip(task_ds, inferences):
        tokens = trim_eos(vocabulary.decode_tf(np.asarray(output, np.int32)))
        start_time = float(first(inp["input_times"]))
        start_time -= start_time % (1 / codec.steps_per_second)
        uid = first(inp["unique_id"])
        uid = uid.decode() if isinstance(uid, bytes) else str(uid)
        if "sequence" in inp:
            any_sequence = True
            seq = inp["sequence"]
            seq = seq if hasattr(seq, "id") or isinstance(seq, (str, bytes)) else first(seq)
            if isinstance(seq, (bytes, str)) and len(seq) == 0:
                seq = None                        # later segments of a track carry an empty string (:85)
            if seq is not None:
                ref_ids[uid] = note_sequence_id(seq)          # (the reference keeps the last one it sees as well, :104-108)
        predictions.append({"unique_id": uid,
                            "est_tokens": tokens, "start_time": start_time,
                            "raw_inputs": inp.get("raw_inputs", [])})
    full = metrics_utils.combine_predictions_by_id(
        predictions, lambda preds: metrics_utils.event_predictions_to_ns(preds, codec=codec,
                                                                         encoding_spec=encoding_spec))
    if any_sequence and sorted(ref_ids.keys()) != sorted(full.keys()):      # the reference asserts it, mt3/inference.py:114
        missing = sorted(set(full) - set(ref_ids))
        extra = sorted(set(ref_ids) - set(full))
        raise MissingSequenceError("write_inferences_to_file: tracks without a NoteSequence in 'sequence': %s; "
                                   "sequences without a track: %s" % (missing[:8], extra[:8]))
    with open(path, "w") as f:
        for uid in sorted(full.keys()):
            notes = [{"start_time": n.start_time, "end_time": n.end_time, "pitch": n.pitch, "velocity": n.velocity,
                      "program": n.program, "is_drum": n.is_drum} for n in full[uid]["est_ns"].notes]
            f.write(json.dumps({"id": ref_ids[uid] if any_sequence else uid, "est_notes": notes}) + "\n")
