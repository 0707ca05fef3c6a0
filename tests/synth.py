"""Seeded additive-synthesis stand-in for rustysynth + SoundFont (absent here):
a random piano roll of 1-6 simultaneous notes, MIDI 33-96, 6 harmonics with 1/h^2 roll-off,
20 ms attack / exponential decay, with the note list as ground truth (BASELINE config 5)."""
import numpy as np


def piano_roll(sr, seconds, seed, max_poly=6):
    rng = np.random.default_rng(seed)
    n = int(sr * seconds)
    x = np.zeros(n, np.float64)
    notes = []  # (midi, start_s, end_s)
    t = 0.0
    while t < seconds - 0.3:
        dur = float(rng.uniform(0.35, 0.9))
        k = int(rng.integers(1, max_poly + 1))
        chord = sorted(set(int(m) for m in rng.integers(40, 90, k)))
        # keep simultaneous notes at least 2 semitones apart (a semitone pair below ~330 Hz is one VQT peak, lib.rs:22-29)
        kept = []
        for m in chord:
            if all(abs(m - q) >= 2 for q in kept):
                kept.append(m)
        for m in kept:
            notes.append((m, t, min(t + dur, seconds)))
        t += dur
    for (m, s, e) in notes:
        i0, i1 = int(s * sr), int(e * sr)
        tt = np.arange(i1 - i0) / sr
        env = np.minimum(tt / 0.02, 1.0) * np.exp(-tt / 0.8) * np.minimum((tt[-1] - tt) / 0.02 + 1e-9, 1.0)
        f0 = 440.0 * 2 ** ((m - 69) / 12.0)
        tone = sum(np.sin(2 * np.pi * f0 * h * tt + 0.3 * h) / h ** 2 for h in range(1, 7) if f0 * h < sr / 2)
        x[i0:i1] += 0.08 * env * tone
    return x.astype(np.float32), notes


def active_notes(notes, t):
    return sorted(m for (m, s, e) in notes if s + 0.12 <= t <= e - 0.05)
