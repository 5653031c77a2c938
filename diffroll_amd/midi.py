"""Roll -> notes -> MIDI, the user-visible product of the reference's sampling driver
(task/diffusion.py:598-618: extract_notes_wo_velocity -> save_midi), SURVEY.md 8f-2.

The note scan runs on the GPU (dr_note_runs); the MIDI file is written by a small dependency-free
writer (the reference uses `mido`, which is not a dependency here).  Timing follows save_midi
(task/diffusion.py:1235-1265): mido's default 480 ticks per beat, ticks_per_second = 2 * 480, note-on /
note-off pairs sorted by time, velocity int(v * 127) capped at 127.
"""
from __future__ import annotations

import struct
from typing import List, Sequence, Tuple

import numpy as np
import torch

MIN_MIDI = 21                 # task/diffusion.py:17
TICKS_PER_BEAT = 480          # mido.MidiFile default
TICKS_PER_SECOND = TICKS_PER_BEAT * 2.0   # task/diffusion.py:1248


def notes_from_runs(note_end: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(T, 88) int array from Engine.note_runs -> (pitches (N,), intervals (N, 2)) in the reference's order
    (np.nonzero: frame-major, then pitch)."""
    frame_locs, pitch_locs = np.nonzero(note_end)
    intervals = np.stack([frame_locs, note_end[frame_locs, pitch_locs]], axis=1).astype(np.int64).reshape(-1, 2)
    return pitch_locs.astype(np.int64), intervals


def extract_notes_wo_velocity(engine, roll: torch.Tensor, threshold: float = 0.5) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Batched counterpart of task/diffusion.py:1185-1233 for the way the drivers call it (onsets == frames
    == roll, one threshold, rule1).  roll (B, T, 88) or (B, 1, T, 88) -> per sample (pitches, intervals)."""
    if roll.dim() == 4:
        roll = roll[:, 0]
    runs = engine.note_runs(roll, threshold).cpu().numpy()
    return [notes_from_runs(runs[b]) for b in range(runs.shape[0])]


def _vlq(n: int) -> bytes:
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def save_midi(path: str, pitches: Sequence[int], intervals_s: Sequence[Sequence[float]],
              velocities: Sequence[float]) -> None:
    """Write a single-track MIDI file.  pitches: MIDI note numbers; intervals_s: (onset, offset) in seconds;
    velocities: as the reference passes them (int(v * 127), capped at 127: it passes 127 -> 127)."""
    events = []
    for i in range(len(pitches)):
        vel = min(int(velocities[i] * 127), 127)
        events.append((float(intervals_s[i][0]), 0x90, int(pitches[i]), vel))
        events.append((float(intervals_s[i][1]), 0x80, int(pitches[i]), vel))
    events.sort(key=lambda ev: ev[0])          # stable, like the reference's list.sort
    track = bytearray()
    last_tick = 0
    for t, status, note, vel in events:
        tick = int(t * TICKS_PER_SECOND)
        track += _vlq(tick - last_tick) + bytes([status, note & 0x7F, vel & 0x7F])
        last_tick = tick
    track += b"\x00\xff\x2f\x00"               # end of track
    with open(path, "wb") as f:
        f.write(b"MThd" + struct.pack(">IHHH", 6, 1, 1, TICKS_PER_BEAT))
        f.write(b"MTrk" + struct.pack(">I", len(track)) + bytes(track))


def read_midi_notes(path: str) -> List[Tuple[int, int, int, int]]:
    """Minimal reader for files written by save_midi: [(tick, status, note, velocity)] (tests only)."""
    data = open(path, "rb").read()
    assert data[:4] == b"MThd" and data[14:18] == b"MTrk"
    n = struct.unpack(">I", data[18:22])[0]
    body, i, tick, out = data[22:22 + n], 0, 0, []
    while i < len(body):
        d = 0
        while True:
            c = body[i]
            i += 1
            d = (d << 7) | (c & 0x7F)
            if not c & 0x80:
                break
        tick += d
        st = body[i]
        if st == 0xFF:
            break
        out.append((tick, st, body[i + 1], body[i + 2]))
        i += 3
    return out


def export_midi(engine, roll: torch.Tensor, path_prefix: str, threshold: float = 0.5, hop_length: int = 512,
                sample_rate: int = 16000, generation_filter: float = 0.0, clean_prefix: str = None) -> List[str]:
    """roll (B,1,T,88) -> per sample `<path_prefix><i>.mid` with every extracted note (the reference's raw_midi_*)
    and, when clean_prefix is given, `<clean_prefix><i>.mid` without the notes not longer than generation_filter
    seconds (its clean_midi_e*), as predict_step does (task/diffusion.py:598-618): frames -> seconds with hop/sr
    (the reference's predict_step uses a stale HOP_LENGTH = 160 constant there, its test_step the model's hop: the
    caller chooses), bins -> MIDI numbers MIN_MIDI + bin, velocity 127.  Returns the raw paths."""
    paths = []
    scaling = hop_length / sample_rate
    for i, (pitches, intervals) in enumerate(extract_notes_wo_velocity(engine, roll, threshold)):
        iv = intervals.astype(np.float64) * scaling
        path = f"{path_prefix}{i}.mid"
        save_midi(path, (MIN_MIDI + pitches).tolist(), iv.tolist(), [127] * len(pitches))
        paths.append(path)
        if clean_prefix is not None:
            keep = (iv[:, 1] - iv[:, 0]) > generation_filter if len(iv) else np.zeros(0, dtype=bool)
            save_midi(f"{clean_prefix}{i}.mid", (MIN_MIDI + pitches[keep]).tolist(), iv[keep].tolist(),
                      [127] * int(keep.sum()))
    return paths
