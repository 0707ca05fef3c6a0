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


def _as_secs_f32_v(ns: np.ndarray) -> np.ndarray:
    return (ns // 1_000_000_000).astype(f32) + (ns % 1_000_000_000).astype(f32) / f32(1_000_000_000.0)


class OracleAnalysisStateVec:
    """The same restatement with the per-bin loops written as NumPy float32 ARRAY operations (every elementwise op rounds to f32
    exactly as the scalar op does; expf / powf are the same glibc calls, over arrays; the sums the reference accumulates bin by
    bin are taken with np.cumsum, which adds sequentially in f32 — not np.sum's pairwise order).  ~25x faster than the scalar
    class above, which stays the literal line-by-line form; tests/test_oracle_analysis_vec.py requires the two to agree BIT FOR BIT
    on every field of every frame, so the GPU tests can run this one over thousands of frames."""

    def __init__(self, min_freq, octaves, bpo, **kw):
        self.min_freq, self.octaves, self.bpo = f32(min_freq), int(octaves), int(bpo)
        self.n = n = self.octaves * self.bpo
        p = dict(peak=(10.0, 4.0), bass=(5.0, 3.5), highest_bassnote=28, base_ns=70_000_000, cmin=0.6, cmax=2.0,
                 note_ns=3_500_000_000, scene_ns=800_000_000, tuning_ns=4_000_000_000, harmonic_threshold=0.3)
        p.update(kw)
        self.p = p
        b = np.arange(n).astype(f32)
        self.mult = (f32(1.5) - f32(0.5) * (b / f32(self.bpo) / f32(self.octaves))).astype(f32)   # analysis.rs:310-313
        base_ms = p["base_ns"] // 1_000_000
        self.h_ns = (f32(base_ms) * self.mult).astype(np.int64) * 1_000_000    # analysis.rs:204-218: the per-bin horizons of new()
        self.h_none = False
        self.sm = np.zeros(n, f32)
        self.peakfiltered = np.zeros(n, f32)
        self.afterglow = np.zeros(n, f32)
        self.peaks = np.zeros(0, np.uint32)
        self.centers = np.zeros(0, f32)
        self.sizes = np.zeros(0, f32)
        self.calm = np.zeros(n, f32)
        self.released = np.zeros(n, f32)
        self.pitch_accuracy = np.zeros(n, f32)
        self.pitch_deviation = np.zeros(n, f32)
        self.scene = f32(0)
        self.tuning = f32(0)
        self.glow_k = (f32(0.85) - f32(0.15) * (b / f32(n))).astype(f32)          # afterglow.rs:27-36

    def update_vqt_smoothing_duration(self, dur_ns):
        self.p["base_ns"] = dur_ns if dur_ns is not None else 0
        if dur_ns is not None:
            self.h_ns = (f32(dur_ns // 1_000_000) * self.mult).astype(np.int64) * 1_000_000
            self.h_none = False
        else:
            self.h_none = True

    @staticmethod
    def _alpha(ts_ns, h_s):
        with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
            arg = f32(f32(-2.0) * _as_secs_f32(ts_ns)) / h_s
        if np.ndim(arg) == 0:
            return f32(1.0) - O.expf(arg)
        return f32(1.0) - O.expf_v(arg)

    def preprocess(self, x, ts_ns):
        x = np.asarray(x, f32)
        assert x.size == self.n
        p, n = self.p, self.n
        cm = f32(p["cmin"]) + f32(f32(p["cmax"]) - f32(p["cmin"])) * self.scene
        base_ms = p["base_ns"] // 1_000_000
        if base_ms > 0:
            dur = (f32(base_ms) * self.mult).astype(f32) * cm
            self.h_ns = dur.astype(np.int64) * 1_000_000            # `duration_ms as u64`: truncation (dur >= 0 here)
            self.h_none = False
        if self.h_none:
            self.sm = x.copy()
        else:
            alpha = self._alpha(ts_ns, _as_secs_f32_v(self.h_ns))
            self.sm = (self.sm + (alpha * (x - self.sm)).astype(f32)).astype(f32)
        sm = self.sm
        ap = O.OracleAnalysisParams(p["peak"][0], p["peak"][1], p["bass"][0], p["bass"][1], p["highest_bassnote"],
                                    p["harmonic_threshold"])
        self.peaks, self.centers, self.sizes = O.analyze_frame(sm, float(self.min_freq), self.octaves, self.bpo, ap)
        mask = np.zeros(n, bool)
        mask[self.peaks] = True
        self.peakfiltered = np.where(mask, sm, f32(0)).astype(f32)
        g = (self.afterglow * self.glow_k).astype(f32)
        self.afterglow = np.where(g < sm, sm, g).astype(f32)
        # calmness.rs:23-95
        radius = self.bpo // 12 // 3
        around = np.zeros(n, bool)
        for pk in O.find_peaks(x, self.bpo, p["peak"][0], p["peak"][1]):
            around[max(0, int(pk) - radius):min(n, int(pk) + radius)] = True
        a_c = self._alpha(ts_ns, _as_secs_f32(p["note_ns"]))
        calm_up = (self.calm + (a_c * (f32(1.0) - self.calm)).astype(f32)).astype(f32)
        calm_dn = (self.calm + (a_c * (f32(0.0) - self.calm)).astype(f32)).astype(f32)
        rel_dn = (self.released + (a_c * (f32(0.0) - self.released)).astype(f32)).astype(f32)
        self.calm = np.where(around, calm_up, calm_dn).astype(f32)
        self.released = np.where(around, calm_up, rel_dn).astype(f32)
        power = O.powf_v(10.0, (sm / f32(10.0)).astype(f32))
        rw = (rel_dn * f32(0.3)).astype(f32)
        contrib = (~around) & (rel_dn > f32(0.01))
        ws = np.where(around, (calm_up * power).astype(f32), np.where(contrib, (rel_dn * rw).astype(f32), f32(0))).astype(f32)
        w = np.where(around, power, np.where(contrib, rw, f32(0))).astype(f32)
        wsum, wt = np.cumsum(ws, dtype=f32)[-1], np.cumsum(w, dtype=f32)[-1]    # sequential f32 sums, bin order (a skipped bin adds an exact 0)
        if wt > 0:
            a_s = self._alpha(ts_ns, _as_secs_f32(p["scene_ns"]))
            self.scene = f32(self.scene + f32(a_s * f32(f32(wsum / wt) - self.scene)))
        # pitch_analysis.rs:12-75
        self.pitch_accuracy = np.zeros(n, f32)
        self.pitch_deviation = np.zeros(n, f32)
        c, s = self.centers, self.sizes
        if c.size:
            power_p = O.powf_v(10.0, (s / f32(10.0)).astype(f32))
            semis = ((c * f32(12.0)).astype(f32) / f32(self.bpo)).astype(f32)
            rnd = (np.floor(np.abs(semis) + f32(0.5)) * np.sign(semis)).astype(f32)   # round half away from zero
            dev = (semis - rnd).astype(f32)
            psum = np.cumsum(power_p, dtype=f32)[-1]
            isum = np.cumsum((np.abs(dev) * power_p).astype(f32), dtype=f32)[-1]
            acc = np.maximum((f32(1.0) - (f32(2.0) * np.abs(dev)).astype(f32)).astype(f32), f32(0))
            bins = np.floor(np.abs(c) + f32(0.5)).astype(np.int64)
            for i in range(c.size):                                   # later peaks overwrite earlier ones
                if bins[i] < n:
                    self.pitch_accuracy[bins[i]] = acc[i]
                    self.pitch_deviation[bins[i]] = dev[i]
            avg = f32(isum / psum) if psum > 0 else f32(0)
        else:
            avg = f32(0)
        a_t = self._alpha(ts_ns, _as_secs_f32(p["tuning_ns"]))
        self.tuning = f32(self.tuning + f32(a_t * f32(f32(f32(100.0) * avg) - self.tuning)))
