"""Note-level evaluation of test_step (task/diffusion.py:385-410): the reference scores the extracted notes with
``mir_eval.transcription.precision_recall_f1_overlap(..., offset_ratio=None)`` (imported as ``evaluate_notes``).

PARITY UNPINNED: mir_eval==0.7 (requirements.txt) is third-party, un-vendored and absent from this image.  This
module restates its published algorithm - onset-only note matching:

    a reference note and an estimated note HIT when |onset_ref - onset_est| (rounded to 6 decimals) <= 50 ms and
    their pitches differ by <= 50 cents; the score counts a MAXIMUM bipartite matching of the hit graph:
    precision = |M| / n_est, recall = |M| / n_ref, F = 2PR / (P + R); empty reference or estimate -> all 0.

The size of a maximum matching is unique, so the metrics do not depend on which matching the library's
Hopcroft-Karp returns.  Host-side integer / graph work on a few hundred notes per clip: numpy, no kernel.
tests/test_host_cpu.py pins it against exhaustive search on small cases.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np

MIN_MIDI = 21            # task/diffusion.py:17
ONSET_TOLERANCE = 0.05   # mir_eval defaults
PITCH_TOLERANCE = 50.0   # cents
_N_DECIMALS = 6


def midi_to_hz(midi):
    """mir_eval.util.midi_to_hz."""
    return 440.0 * (2.0 ** ((np.asarray(midi, dtype=np.float64) - 69.0) / 12.0))


def _max_matching(adj: Dict[int, List[int]]) -> int:
    """Size of a maximum bipartite matching (Kuhn's augmenting paths; graphs here are tiny and sparse)."""
    match_ref: Dict[int, int] = {}

    def augment(est: int, seen: set) -> bool:
        for ref in adj[est]:
            if ref in seen:
                continue
            seen.add(ref)
            if ref not in match_ref or augment(match_ref[ref], seen):
                match_ref[ref] = est
                return True
        return False

    size = 0
    for est in adj:
        if augment(est, set()):
            size += 1
    return size


def match_count(ref_intervals, ref_pitches_hz, est_intervals, est_pitches_hz,
                onset_tolerance: float = ONSET_TOLERANCE, pitch_tolerance: float = PITCH_TOLERANCE) -> int:
    ref_intervals = np.asarray(ref_intervals, dtype=np.float64).reshape(-1, 2)
    est_intervals = np.asarray(est_intervals, dtype=np.float64).reshape(-1, 2)
    if len(ref_intervals) == 0 or len(est_intervals) == 0:
        return 0
    onset = np.around(np.abs(np.subtract.outer(ref_intervals[:, 0], est_intervals[:, 0])), decimals=_N_DECIMALS)
    cents = np.abs(1200.0 * np.subtract.outer(np.log2(np.asarray(ref_pitches_hz, dtype=np.float64)),
                                              np.log2(np.asarray(est_pitches_hz, dtype=np.float64))))
    hits = np.where((onset <= onset_tolerance) & (cents <= pitch_tolerance))
    adj: Dict[int, List[int]] = {}
    for ref_i, est_i in zip(*hits):
        adj.setdefault(int(est_i), []).append(int(ref_i))
    return _max_matching(adj)


def evaluate_notes(ref_intervals, ref_pitches_hz, est_intervals, est_pitches_hz) -> Tuple[float, float, float]:
    """(precision, recall, F1) of onset-only note matching, as the reference's
    ``evaluate_notes(i_ref, p_ref, i_est, p_est, offset_ratio=None)`` (task/diffusion.py:410)."""
    n_ref, n_est = len(ref_pitches_hz), len(est_pitches_hz)
    if n_ref == 0 or n_est == 0:
        return 0.0, 0.0, 0.0
    m = match_count(ref_intervals, ref_pitches_hz, est_intervals, est_pitches_hz)
    p, r = m / n_est, m / n_ref
    f = 0.0 if p + r == 0 else 2 * p * r / (p + r)
    return p, r, f


def note_scores(ref_notes: Sequence[Tuple[np.ndarray, np.ndarray]], est_notes: Sequence[Tuple[np.ndarray, np.ndarray]],
                hop_length: int, sample_rate: int) -> List[Tuple[float, float, float]]:
    """Per-sample (P, R, F1) from (pitches, frame intervals) pairs as extract_notes_wo_velocity returns them:
    frames -> seconds with hop_length / sample_rate, key index -> Hz with midi_to_hz(21 + key) (:400-408)."""
    scaling = hop_length / sample_rate
    out = []
    for (p_ref, i_ref), (p_est, i_est) in zip(ref_notes, est_notes):
        i_ref_s = (np.asarray(i_ref, dtype=np.float64) * scaling).reshape(-1, 2)
        i_est_s = (np.asarray(i_est, dtype=np.float64) * scaling).reshape(-1, 2)
        out.append(evaluate_notes(i_ref_s, midi_to_hz(MIN_MIDI + np.asarray(p_ref)), i_est_s,
                                  midi_to_hz(MIN_MIDI + np.asarray(p_est))))
    return out
