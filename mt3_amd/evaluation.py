"""Note-level comparison of engine precisions on ONE file (BASELINE.json north_star: "decoded note onsets/offsets within a
stated fp tolerance"; SURVEY.md 8(d): onset +-50 ms, offset max(50 ms, 20 %), the mir_eval rule of mt3/metrics.py:255-290).

The f32 engine is the reference's own precision and token-exact against the oracle (tests/test_gpu_parity_*.py,
bench.py cpu_baseline.parity), so ITS notes stand for the reference's; every other engine configuration -- bf16 operands,
e4m3 K/V caches, MXFP8 encoder -- is scored against them on the same samples through the same drop-in class
(`InferenceModel`, NB:283-308).  Used by bench.py (`extra.divergence_vs_f32.<mode>.{boosted,trained}`) and by
tests/test_gpu_note_tolerance.py; nothing here is on the product path.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np

from . import metrics
from . import metrics_utils
from . import network

# engine configurations bench.py times next to the f32 headline: key -> (dtype, kv_dtype, dense_dtype)
REDUCED_MODES = {
    "bf16": ("bfloat16", "", ""),
    "fp8_kv": ("bfloat16", "fp8_e4m3", ""),
    "fp8_kv_mx8": ("bfloat16", "fp8_e4m3", "fp8_e4m3"),
}


def file_notes_and_tokens(model, audio) -> Tuple[Any, np.ndarray]:
    """`InferenceModel.__call__` (NB:283-308) that also hands back the token rows: (NoteSequence, int32 [segments, 1024])"""
    examples = model.preprocess(model.audio_to_dataset(audio), host_inputs=False)
    batch, model._logmel_dev = {"encoder_input_tokens": model._logmel_dev}, None
    tokens = model.predict_tokens(batch)
    preds = [model.postprocess(t, ex) for t, ex in zip(tokens, examples)]
    ns = metrics_utils.event_predictions_to_ns(preds, codec=model.codec, encoding_spec=model.encoding_spec)["est_ns"]
    return ns, tokens


def note_divergence(ref_ns, est_ns, ref_tokens: Optional[np.ndarray] = None,
                    est_tokens: Optional[np.ndarray] = None) -> Dict[str, Any]:
    """est against ref: note counts, onset / onset + offset precision-recall-F1 under the reference's rule (pitches as
    note numbers, mt3/metrics.py:267-290) and with exact pitch ("hz"), and -- with the token rows -- how the streams
    themselves differ."""
    out: Dict[str, Any] = {"ref_notes": len(ref_ns.notes), "est_notes": len(est_ns.notes)}
    # mir_eval.transcription.validate rejects pitch 0 (not a positive frequency) and the reference's call would raise on
    # it (mt3/metrics.py:267-290); random-init streams do contain MIDI pitch 0, so those notes are set aside and counted
    pitch0 = sum(n.pitch <= 0 for n in ref_ns.notes) + sum(n.pitch <= 0 for n in est_ns.notes)
    if pitch0:
        from .note_sequences import NoteSequence
        out["pitch_0_notes_set_aside"] = int(pitch0)
        ref_ns = NoteSequence(notes=[n for n in ref_ns.notes if n.pitch > 0], total_time=ref_ns.total_time)
        est_ns = NoteSequence(notes=[n for n in est_ns.notes if n.pitch > 0], total_time=est_ns.total_time)
    same = [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in ref_ns.notes] == \
           [(n.start_time, n.end_time, n.pitch, n.velocity, n.program, n.is_drum) for n in est_ns.notes]
    out["notes_identical"] = bool(same)
    for unit in ("note_number", "hz"):
        sc = metrics.transcription_scores(ref_ns, est_ns, pitch_unit=unit)
        out["onset_f1_" + unit] = sc["Onset F1"]
        out["onset_offset_f1_" + unit] = sc["Onset + offset F1"]
        if unit == "note_number":
            out["onset_precision"], out["onset_recall"] = sc["Onset precision"], sc["Onset recall"]
    if ref_tokens is not None and est_tokens is not None:
        t = metrics.token_stream_divergence(ref_tokens, est_tokens)
        for k in ("rows", "identical_rows_frac", "equal_positions_frac", "median_first_divergence_step"):
            out[k] = t[k]
        ref_len = np.where((ref_tokens == -1).any(1), (ref_tokens == -1).argmax(1), ref_tokens.shape[1])
        out["mean_tokens_per_row"] = float(ref_len.mean())
    return out


def compare_engines(params, audio, shape: network.T5Config = network.MT3_SMALL,
                    modes: Sequence[str] = tuple(REDUCED_MODES), model_type: str = "mt3", decoding: str = "beam1",
                    max_slots: int = 256, truth=None) -> Dict[str, Any]:
    """One file through `InferenceModel` once per engine configuration: the f32 engine first (the reference's precision),
    then every mode of `modes`, each scored against the f32 engine's notes.  truth: optional NoteSequence the audio was
    synthesised from -- every engine (f32 included) is then also scored against IT ("accuracy", under `vs_truth`)."""
    from . import inference
    out: Dict[str, Any] = {}

    def run(dtype, kv, dense):
        cfg = dataclasses.replace(shape, kv_dtype=kv, dense_dtype=dense)
        m = inference.InferenceModel(params, model_type, config=cfg, dtype=dtype, decoding=decoding, max_slots=max_slots)
        ns, tok = file_notes_and_tokens(m, audio)
        del m
        return ns, tok

    ref_ns, ref_tok = run("float32", "", "")
    out["f32"] = {"notes": len(ref_ns.notes), "segments": int(ref_tok.shape[0])}
    if truth is not None:
        out["f32"]["vs_truth"] = note_divergence(truth, ref_ns)
    for key in modes:
        dtype, kv, dense = REDUCED_MODES[key]
        try:
            ns, tok = run(dtype, kv, dense)
            out[key] = note_divergence(ref_ns, ns, ref_tok, tok)
            if truth is not None:
                out[key]["vs_truth"] = note_divergence(truth, ns)
        except Exception as ex:                       # a report: one mode failing must not hide the others
            out[key] = {"error": repr(ex)[:300]}
    return out
