"""Oracle restatement of the stateful AnalysisState (TEST INFRASTRUCTURE, not product code).

Follows pitchvis_analysis/src/analysis.rs:192-404, analysis_modules/afterglow.rs:10-36,
calmness.rs:23-95, pitch_analysis.rs:12-75 and util.rs:91-137 in NumPy float32 scalar
arithmetic (every intermediate is rounded to f32, in the reference's operation order); the peak
functions are the C oracle's (oracle/pvq_oracle.c).  Durations are integer nanoseconds like
std::time::Duration."""
from __future__ import annotations

import numpy as np

import oracle as O

f32 = np.float32


def _as_secs_f32(ns: int) -> np.float32:
    return f32(ns // 1_000_000_000) + f32(ns % 1_000_000_000) / f32(1_000_000_000.0)


class Ema:
    """util.rs:91-137"""

    def __init__(self, horizon_ns, value):
        self.h = horizon_ns  # None = no smoothing
        self.y = f32(value)

    def update(self, new, ts_ns):
        new = f32(new)
        if self.h is not None:
            with np.errstate(divide="ignore", over="ignore"):
                alpha = f32(1.0) - O.expf(f32(f32(-2.0) * _as_secs_f32(ts_ns)) / _as_secs_f32(self.h))
            self.y = f32(self.y + f32(alpha * f32(new - self.y)))
        else:
            self.y = new

    def copy(self):
        e = Ema(self.h, self.y)
        return e


class OracleAnalysisState:
    def __init__(self, min_freq, octaves, bpo, **kw):
        self.min_freq, self.octaves, self.bpo = f32(min_freq), int(octaves), int(bpo)
        self.n = self.octaves * self.bpo
        p = dict(peak=(10.0, 4.0), bass=(5.0, 3.5), highest_bassnote=28, base_ns=70_000_000, cmin=0.6, cmax=2.0,
                 note_ns=3_500_000_000, scene_ns=800_000_000, tuning_ns=4_000_000_000, harmonic_threshold=0.3)
        p.update(kw)
        self.p = p
        base_ms = p["base_ns"] // 1_000_000
        self.smoothed = []
        for b in range(self.n):
            frac = f32(b) / f32(self.bpo) / f32(self.octaves)
            mult = f32(1.5) - f32(0.5) * frac
            dur_ms = int(f32(base_ms) * mult)
            self.smoothed.append(Ema(dur_ms * 1_000_000, 0.0))
        self.peakfiltered = np.zeros(self.n, f32)
        self.afterglow = np.zeros(self.n, f32)
        self.peaks = np.zeros(0, np.uint32)
        self.centers = np.zeros(0, f32)
        self.sizes = np.zeros(0, f32)
        self.calm = [Ema(p["note_ns"], 0.0) for _ in range(self.n)]
        self.released = [Ema(p["note_ns"], 0.0) for _ in range(self.n)]
        self.pitch_accuracy = np.zeros(self.n, f32)
        self.pitch_deviation = np.zeros(self.n, f32)
        self.scene = Ema(p["scene_ns"], 0.0)
        self.tuning = Ema(p["tuning_ns"], 0.0)

    def update_vqt_smoothing_duration(self, dur_ns):
        self.p["base_ns"] = dur_ns if dur_ns is not None else 0
        for b, e in enumerate(self.smoothed):
            if dur_ns is not None:
                frac = f32(b) / f32(self.bpo) / f32(self.octaves)
                mult = f32(1.5) - f32(0.5) * frac
                e.h = int(f32(dur_ns // 1_000_000) * mult) * 1_000_000
            else:
                e.h = None

    def preprocess(self, x, ts_ns):
        x = np.asarray(x, f32)
        assert x.size == self.n
        p = self.p
        cm = f32(p["cmin"]) + f32(f32(p["cmax"]) - f32(p["cmin"])) * self.scene.y
        base_ms = p["base_ns"] // 1_000_000
        for b, e in enumerate(self.smoothed):
            if base_ms > 0:
                frac = f32(b) / f32(self.bpo) / f32(self.octaves)
                mult = f32(1.5) - f32(0.5) * frac
                dur = f32(f32(base_ms) * mult) * cm
                e.h = int(dur) * 1_000_000
            e.update(x[b], ts_ns)
        sm = np.array([e.y for e in self.smoothed], f32)
        ap = O.OracleAnalysisParams(p["peak"][0], p["peak"][1], p["bass"][0], p["bass"][1], p["highest_bassnote"],
                                    p["harmonic_threshold"])
        self.peaks, self.centers, self.sizes = O.analyze_frame(sm, float(self.min_freq), self.octaves, self.bpo, ap)
        mask = np.zeros(self.n, bool)
        mask[self.peaks] = True
        self.peakfiltered = np.where(mask, sm, f32(0)).astype(f32)
        for i in range(self.n):
            g = f32(self.afterglow[i] * f32(f32(0.85) - f32(0.15) * f32(f32(i) / f32(self.n))))
            self.afterglow[i] = sm[i] if g < sm[i] else g
        # calmness.rs
        radius = self.bpo // 12 // 3
        around = np.zeros(self.n, bool)
        for pk in O.find_peaks(x, self.bpo, p["peak"][0], p["peak"][1]):
            around[max(0, int(pk) - radius):min(self.n, int(pk) + radius)] = True
        wsum, wt = f32(0), f32(0)
        for b in range(self.n):
            if around[b]:
                self.calm[b].update(1.0, ts_ns)
                self.released[b] = self.calm[b].copy()
                power = O.powf(10.0, f32(sm[b] / f32(10.0)))
                wsum = f32(wsum + f32(self.calm[b].y * power))
                wt = f32(wt + power)
            else:
                self.calm[b].update(0.0, ts_ns)
                self.released[b].update(0.0, ts_ns)
                rel = self.released[b].y
                if rel > f32(0.01):
                    w = f32(rel * f32(0.3))
                    wsum = f32(wsum + f32(rel * w))
                    wt = f32(wt + w)
        if wt > 0:
            self.scene.update(f32(wsum / wt), ts_ns)
        # pitch_analysis.rs
        isum, psum = f32(0), f32(0)
        self.pitch_accuracy[:] = 0
        self.pitch_deviation[:] = 0
        for c, s in zip(self.centers, self.sizes):
            power = O.powf(10.0, f32(s / f32(10.0)))
            psum = f32(psum + power)
            semis = f32(f32(c * f32(12.0)) / f32(self.bpo))
            rnd = f32(np.floor(np.abs(semis) + f32(0.5)) * np.sign(semis))  # round half away from zero
            dev = f32(semis - rnd)
            isum = f32(isum + f32(np.abs(dev) * power))
            bin_idx = int(np.floor(np.abs(c) + f32(0.5)))
            if bin_idx < self.n:
                self.pitch_accuracy[bin_idx] = max(f32(f32(1.0) - f32(f32(2.0) * np.abs(dev))), f32(0))
                self.pitch_deviation[bin_idx] = dev
        avg = f32(isum / psum) if psum > 0 else f32(0)
        self.tuning.update(f32(f32(100.0) * avg), ts_ns)
