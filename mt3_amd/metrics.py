"""Note-level transcription scores: the matching rule the reference delegates to
`mir_eval.transcription.precision_recall_f1_overlap` (mt3/metrics.py:255-319): a reference note and
an estimated note match if their onsets are within 50 ms, their pitches are equal (mir_eval's 50-cent
tolerance on MIDI notes) and -- unless offsets are ignored -- their offsets are within
max(50 ms, 20 % of the reference duration); the score uses a MAXIMUM bipartite matching.
mir_eval is not installable here: PARITY UNPINNED against it (checked against brute force in tests).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .note_sequences import NoteSequence


def _arrays(ns: NoteSequence, drums: bool = False):
    notes = [n for n in ns.notes if bool(n.is_drum) == drums]
    iv = np.array([[n.start_time, n.end_time] for n in notes], np.float64).reshape(-1, 2)
    return iv, np.array([n.pitch for n in notes], np.int64), np.array([n.program for n in notes], np.int64)


def match_notes(ref_iv, ref_pitch, est_iv, est_pitch, onset_tolerance=0.05, offset_ratio: Optional[float] = 0.2,
                offset_min_tolerance=0.05, ref_prog=None, est_prog=None) -> int:
    """Size of the maximum matching between reference and estimated notes."""
    if len(ref_iv) == 0 or len(est_iv) == 0:
        return 0
    ok = np.abs(ref_iv[:, None, 0] - est_iv[None, :, 0]) <= onset_tolerance
    ok &= ref_pitch[:, None] == est_pitch[None, :]
    if offset_ratio is not None:
        tol = np.maximum(offset_min_tolerance, offset_ratio * (ref_iv[:, 1] - ref_iv[:, 0]))
        ok &= np.abs(ref_iv[:, None, 1] - est_iv[None, :, 1]) <= tol[:, None]
    if ref_prog is not None:
        ok &= ref_prog[:, None] == est_prog[None, :]
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    m = maximum_bipartite_matching(csr_matrix(ok.astype(np.int8)), perm_type="column")
    return int((m >= 0).sum())


def _prf(matched, n_ref, n_est):
    p = matched / n_est if n_est else 0.0
    r = matched / n_ref if n_ref else 0.0
    return p, r, (2 * p * r / (p + r) if p + r else 0.0)


def transcription_scores(ref_ns: NoteSequence, est_ns: NoteSequence, use_programs: bool = False) -> Dict[str, float]:
    """Onset-only and onset+offset precision / recall / F1 (non-drum notes), as in metrics.py:255-319."""
    ri, rp, rg = _arrays(ref_ns)
    ei, ep, eg = _arrays(est_ns)
    kw = dict(ref_prog=rg, est_prog=eg) if use_programs else {}
    out = {}
    for name, ratio in (("Onset", None), ("Onset + offset", 0.2)):
        p, r, f = _prf(match_notes(ri, rp, ei, ep, offset_ratio=ratio, **kw), len(ri), len(ei))
        out[name + " precision"], out[name + " recall"], out[name + " F1"] = p, r, f
    return out
