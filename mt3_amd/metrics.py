"""Note-level transcription scores and the precision-mode divergence report.

`precision_recall_f1_overlap` restates the rule the reference delegates to
`mir_eval.transcription.precision_recall_f1_overlap` (called at mt3/metrics.py:97-99, 163-166, 267-290), from
mir_eval's documented algorithm (`match_notes`): with D = |difference| ROUNDED TO 6 DECIMALS (mir_eval rounds so
that a distance of exactly the tolerance is not lost to float noise) and comparisons NON-strict (`<=`; `strict=True`
switches every one of them to `<`):
  onset   D(ref_onset, est_onset)   <= onset_tolerance                                   (0.05 s)
  pitch   1200 |log2(ref_pitch / est_pitch)| <= pitch_tolerance                          (50 cents, not rounded)
  offset  D(ref_offset, est_offset) <= max(offset_ratio * ref_duration, offset_min_tolerance)   (20 %, 0.05 s;
          skipped when offset_ratio is None)
a reference and an estimated note can be paired when all hold; the score counts a MAXIMUM bipartite matching
(mir_eval: Hopcroft-Karp; any maximum matching has the same size); precision = matched / n_est, recall = matched /
n_ref, F = (1 + b^2) P R / (b^2 P + R); everything 0 when either side has no notes.
How the reference CALLS it (mt3/metrics.py:255-290): intervals and pitches come from
`note_seq.sequences_lib.sequence_to_valued_intervals`, which drops zero-length notes and returns MIDI NOTE NUMBERS
[from memory of note_seq; magenta's own onsets-and-frames code converts them with `pretty_midi.note_number_to_hz`
before calling mir_eval, mt3/metrics.py does not] -- so in the reference's figures the "50 cents" act on note numbers:
neighbouring numbers p, p+1 differ by 1200 log2((p+1)/p) cents, i.e. they PASS the pitch test for p >= 35.
`transcription_scores(pitch_unit="note_number")` reproduces that call; `pitch_unit="hz"` is the conventional
metric (exact pitch).  mir_eval and note_seq are not installable here: PARITY UNPINNED against the libraries
themselves; the boundary behaviour above is pinned by tests/test_io_and_metrics.py.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from .note_sequences import NoteSequence

N_DECIMALS = 6          # mir_eval.transcription: precision the onset / offset distances are rounded to


def sequence_to_valued_intervals(ns: NoteSequence, drums: Optional[bool] = None):
    """note_seq.sequences_lib.sequence_to_valued_intervals as mt3/metrics.py uses it: (intervals [n, 2], note numbers,
    velocities), zero-length notes dropped (mir_eval rejects them).  `drums`: None = every note, else filter."""
    notes = [n for n in ns.notes if n.end_time != n.start_time and (drums is None or bool(n.is_drum) == drums)]
    iv = np.array([[n.start_time, n.end_time] for n in notes], np.float64).reshape(-1, 2)
    return iv, np.array([n.pitch for n in notes], np.float64), np.array([n.velocity for n in notes], np.int64)


def _hit_pairs(ref_iv, ref_pitch, est_iv, est_pitch, onset_tolerance, pitch_tolerance, offset_ratio,
               offset_min_tolerance, strict):
    """(ref index, est index) of every pair that passes the rule.  mir_eval builds the dense n_ref x n_est matrices;
    here only pairs whose onsets lie within the tolerance (+ the rounding margin) are ever formed -- the same set,
    without the quadratic memory (a 9-minute file has tens of thousands of notes)."""
    cmp = np.less if strict else np.less_equal
    order = np.argsort(est_iv[:, 0], kind="stable")
    est_on = est_iv[order, 0]
    margin = onset_tolerance + 10.0 ** -N_DECIMALS
    lo = np.searchsorted(est_on, ref_iv[:, 0] - margin, side="left")
    hi = np.searchsorted(est_on, ref_iv[:, 0] + margin, side="right")
    cnt = hi - lo
    ri = np.repeat(np.arange(len(ref_iv)), cnt)
    ei = order[np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)])] if cnt.sum() else np.zeros(0, np.int64)
    onset_d = np.around(np.abs(ref_iv[ri, 0] - est_iv[ei, 0]), decimals=N_DECIMALS)
    ok = cmp(onset_d, onset_tolerance)
    with np.errstate(divide="ignore", invalid="ignore"):
        pitch_d = np.abs(1200.0 * (np.log2(ref_pitch[ri]) - np.log2(est_pitch[ei])))
    ok &= cmp(np.nan_to_num(pitch_d, nan=0.0), pitch_tolerance)      # (log2(0) - log2(0): note number 0 on both sides)
    if offset_ratio is not None:
        offset_d = np.around(np.abs(ref_iv[ri, 1] - est_iv[ei, 1]), decimals=N_DECIMALS)
        tol = np.maximum(offset_ratio * (ref_iv[:, 1] - ref_iv[:, 0]), offset_min_tolerance)
        ok &= cmp(offset_d, tol[ri])
    return ri[ok], ei[ok]


def match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05, pitch_tolerance=50.0,
                offset_ratio: Optional[float] = 0.2, offset_min_tolerance=0.05, strict=False,
                ref_extra=None, est_extra=None) -> int:
    """Size of the maximum matching (mir_eval.transcription.match_notes returns the pairs; only their number enters the
    scores).  ref_extra / est_extra: optional labels that must also be equal (programs)."""
    ref_iv, est_iv = np.asarray(ref_intervals, np.float64).reshape(-1, 2), np.asarray(est_intervals, np.float64).reshape(-1, 2)
    if len(ref_iv) == 0 or len(est_iv) == 0:
        return 0
    ri, ei = _hit_pairs(ref_iv, np.asarray(ref_pitches, np.float64), est_iv, np.asarray(est_pitches, np.float64),
                        onset_tolerance, pitch_tolerance, offset_ratio, offset_min_tolerance, strict)
    if ref_extra is not None:
        keep = np.asarray(ref_extra)[ri] == np.asarray(est_extra)[ei]
        ri, ei = ri[keep], ei[keep]
    if len(ri) == 0:
        return 0
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    g = csr_matrix((np.ones(len(ri), np.int8), (ri, ei)), shape=(len(ref_iv), len(est_iv)))
    m = maximum_bipartite_matching(g, perm_type="column")
    return int((m >= 0).sum())


def f_measure(precision, recall, beta=1.0):
    """mir_eval.util.f_measure"""
    if precision == 0 and recall == 0:
        return 0.0
    return (1 + beta ** 2) * precision * recall / ((beta ** 2) * precision + recall)


def precision_recall_f1_overlap(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance=0.05,
                                pitch_tolerance=50.0, offset_ratio: Optional[float] = 0.2, offset_min_tolerance=0.05,
                                strict=False, beta=1.0):
    """(precision, recall, F) of mir_eval.transcription.precision_recall_f1_overlap (its 4th value, the average overlap
    ratio of the matched pairs, is discarded at every call site of mt3/metrics.py and not computed here)."""
    n_ref, n_est = len(np.asarray(ref_pitches)), len(np.asarray(est_pitches))
    # mir_eval.transcription.validate runs FIRST (before the emptiness check) and, per side, util.validate_frequencies
    # rejects pitches that are not positive -- so a pitch 0 on one side raises even when the other side is empty
    # (an empty side itself only warns there)
    for side in (ref_pitches, est_pitches):
        a = np.asarray(side, np.float64)
        if a.size and (a <= 0).any():
            raise ValueError("precision_recall_f1_overlap: pitches must be positive (mir_eval.transcription.validate)")
    if n_ref == 0 or n_est == 0:
        return 0.0, 0.0, 0.0
    m = match_notes(ref_intervals, ref_pitches, est_intervals, est_pitches, onset_tolerance, pitch_tolerance,
                    offset_ratio, offset_min_tolerance, strict)
    p, r = m / n_est, m / n_ref
    return p, r, f_measure(p, r, beta)


def _hz(note_numbers):
    return 440.0 * 2.0 ** ((np.asarray(note_numbers, np.float64) - 69.0) / 12.0)


def transcription_scores(ref_ns: NoteSequence, est_ns: NoteSequence, pitch_unit: str = "note_number") -> Dict[str, float]:
    """'Onset' and 'Onset + offset' precision / recall / F1 over the non-drum notes, as mt3/metrics.py:228-290 computes
    them for the whole NoteSequence (remove_drums, then the two mir_eval calls).  pitch_unit: "note_number" = the
    reference's own call (see the module docstring), "hz" = note numbers converted first (exact-pitch matching)."""
    if pitch_unit not in ("note_number", "hz"):
        raise ValueError("pitch_unit must be 'note_number' or 'hz'")
    ri, rp, _ = sequence_to_valued_intervals(ref_ns, drums=False)
    ei, ep, _ = sequence_to_valued_intervals(est_ns, drums=False)
    if pitch_unit == "hz":
        rp, ep = _hz(rp), _hz(ep)
    out = {}
    for name, ratio in (("Onset", None), ("Onset + offset", 0.2)):
        p, r, f = precision_recall_f1_overlap(ri, rp, ei, ep, offset_ratio=ratio)
        out[name + " precision"], out[name + " recall"], out[name + " F1"] = p, r, f
    return out


def program_aware_note_scores(ref_ns: NoteSequence, est_ns: NoteSequence, granularity_type: str = "full") -> Dict[str, float]:
    """mt3/metrics.py:35-147 `_program_aware_note_scores`: non-drum programs mapped by the granularity's `program_map_fn`,
    then per (program, is_drum) track (`note_sequences.extract_track`) one `precision_recall_f1_overlap` call -- onsets +
    offsets for non-drum tracks, onsets only for drum tracks (`offset_ratio=None`) -- and precision / recall averaged over
    tracks weighted by their numbers of estimated / reference notes; F from the averaged pair.  Pitches go to the 50-cent
    rule as note NUMBERS, as everywhere in the reference (module docstring).  A track that one side lacks scores 0 there
    (mir_eval returns zeros for an empty side)."""
    from . import vocabularies
    pmap = vocabularies.PROGRAM_MAP_FNS[granularity_type]

    def tracks(ns):
        out = {}
        for n in ns.notes:
            key = (n.program if n.is_drum else pmap(n.program), bool(n.is_drum))
            out.setdefault(key, []).append(n)
        return out
    ref_t, est_t = tracks(ref_ns), tracks(est_ns)
    sums = {True: [0.0, 0, 0.0, 0], False: [0.0, 0, 0.0, 0]}          # is_drum -> [P sum, P count, R sum, R count]
    for key in set(ref_t) | set(est_t):
        is_drum = key[1]
        ri, rp, _ = sequence_to_valued_intervals(NoteSequence(notes=ref_t.get(key, [])))
        ei, ep, _ = sequence_to_valued_intervals(NoteSequence(notes=est_t.get(key, [])))
        p, r, _ = precision_recall_f1_overlap(ri, rp, ei, ep, **({"offset_ratio": None} if is_drum else {}))
        acc = sums[is_drum]
        acc[0] += p * len(ei)
        acc[1] += len(ei)
        acc[2] += r * len(ri)
        acc[3] += len(ri)

    def prf(ps, pc, rs, rc):
        p, r = (ps / pc) if pc else 0, (rs / rc) if rc else 0
        return p, r, f_measure(p, r)
    d, nd = sums[True], sums[False]
    g = granularity_type
    out = {}
    for name, (p, r, f) in (("Onset + offset + program", prf(d[0] + nd[0], d[1] + nd[1], d[2] + nd[2], d[3] + nd[3])),
                            ("Drum onset", prf(*d)), ("Nondrum onset + offset + program", prf(*nd))):
        out[f"{name} precision ({g})"], out[f"{name} recall ({g})"], out[f"{name} F1 ({g})"] = p, r, f
    return out


# ------------------------------------------------------------------------------- precision-mode divergence report
def token_stream_divergence(ref_tokens: np.ndarray, est_tokens: np.ndarray, codec=None, encoding_spec=None,
                            segment_seconds: float = 2.048) -> Dict[str, float]:
    """How far a reduced-precision engine's FREE-RUNNING decode drifts from the f32 engine's on the same inputs
    (VERDICT r2 #1c).  ref_tokens / est_tokens: int32 [B, L] rows after `decode_tf` (-1 from EOS on).  Reports the
    fraction of rows that are identical, the first-divergence step of the others (median, quartiles), the fraction of
    equal positions, and -- with a codec -- note-level F1 of the notes the two streams decode to (the rows taken as
    consecutive segments of one track; `transcription_scores`, both pitch conventions).  With random-init weights one
    flipped arg-max re-rolls the rest of a row, so these figures are an upper bound on what trained weights (peaked
    distributions) would show."""
    a, b = np.asarray(ref_tokens), np.asarray(est_tokens)
    if a.shape != b.shape or a.ndim != 2:
        raise ValueError("token arrays must both be [B, L]")
    neq = a != b
    same_row = ~neq.any(1)
    first = np.where(same_row, a.shape[1], neq.argmax(1))
    out = {"rows": int(a.shape[0]), "steps": int(a.shape[1]), "identical_rows_frac": float(same_row.mean()),
           "equal_positions_frac": float(1.0 - neq.mean()),
           "median_first_divergence_step": float(np.median(first[~same_row])) if (~same_row).any() else None,
           "first_divergence_quartiles": [float(q) for q in np.percentile(first[~same_row], (25, 50, 75))]
           if (~same_row).any() else None,
           "mean_common_prefix_frac": float(first.mean() / a.shape[1])}
    if codec is not None:
        from . import metrics_utils
        from . import note_sequences as NS
        spec = encoding_spec or NS.NoteEncodingWithTiesSpec

        def notes(tok):
            preds = []
            for i, row in enumerate(tok):
                eos = np.nonzero(row == -1)[0]
                t = i * segment_seconds
                preds.append({"est_tokens": row[: eos[0]] if eos.size else row, "start_time": t - t % 0.01})
            return metrics_utils.event_predictions_to_ns(preds, codec, spec)["est_ns"]
        ref_ns, est_ns = notes(a), notes(b)
        out["ref_notes"], out["est_notes"] = len(ref_ns.notes), len(est_ns.notes)
        for unit in ("note_number", "hz"):
            sc = transcription_scores(ref_ns, est_ns, pitch_unit=unit)
            out["onset_f1_" + unit] = sc["Onset F1"]
            out["onset_offset_f1_" + unit] = sc["Onset + offset F1"]
    return out
