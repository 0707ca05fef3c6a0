"""oracle/analysis_state.py holds the AnalysisState restatement twice: the literal scalar form (one Ema object per bin, the
reference's loops line by line: analysis.rs:288-404, calmness.rs:23-95, pitch_analysis.rs:12-75, afterglow.rs) and an array form
that the GPU tests can afford over thousands of frames.  The array form is only trusted because this test requires it to equal
the scalar one BIT FOR BIT on every pub field of every frame, in the three smoothing modes, with a constant and a jittered
frame time, at 36 and 84 bins per octave."""
import numpy as np
import pytest

from oracle.analysis_state import OracleAnalysisState, OracleAnalysisStateVec


def frames(n_frames, n_bins, seed):
    """dB-like frames: a noise floor, notes that start, hold, glide and stop, a silent stretch, a loud stretch"""
    rng = np.random.default_rng(seed)
    x = (rng.random((n_frames, n_bins), dtype=np.float32) * 6.0).astype(np.float32)
    for _ in range(5):
        b0 = int(rng.integers(3, n_bins - 3))
        t0 = int(rng.integers(0, max(1, n_frames - 30)))
        t1 = min(n_frames, t0 + int(rng.integers(20, 120)))
        lvl = float(rng.uniform(18.0, 50.0))
        for t in range(t0, t1):
            b = min(max(b0 + (t - t0) // 37, 2), n_bins - 3)
            x[t, b] = lvl + 0.3 * np.sin(t / 7.0)
            x[t, b - 1] = max(x[t, b - 1], lvl - 9.0)
            x[t, b + 1] = max(x[t, b + 1], lvl - 11.0)
    q = n_frames // 2
    x[q:q + 12] = 0.0
    return x


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("bpo,octaves,mode,jitter", [(36, 7, "default", False), (36, 7, "default", True), (84, 3, "default", False),
                                                      (36, 5, "none", False), (36, 7, "retuned", True), (36, 4, "zero_base", False),
                                                      (36, 7, "nondefault", True), (84, 3, "nondefault", False)])
def test_array_form_equals_scalar_form_bit_for_bit(bpo, octaves, mode, jitter):
    n = bpo * octaves
    kw = dict(base_ns=0) if mode == "zero_base" else {}
    if mode == "nondefault":   # every AnalysisParameters field off its default (the GPU batch's non-default test leans on the array form here too)
        kw = dict(peak=(8.0, 3.0), bass=(4.0, 2.5), highest_bassnote=20, base_ns=40_000_000, cmin=0.5, cmax=3.0, note_ns=2_000_000_000,
                  scene_ns=500_000_000, tuning_ns=3_000_000_000, harmonic_threshold=0.2)
    a, b = OracleAnalysisState(55.0, octaves, bpo, **kw), OracleAnalysisStateVec(55.0, octaves, bpo, **kw)
    if mode == "none":
        a.update_vqt_smoothing_duration(None); b.update_vqt_smoothing_duration(None)
    elif mode == "retuned":
        a.update_vqt_smoothing_duration(120_000_000); b.update_vqt_smoothing_duration(120_000_000)
    x = frames(140, n, 11 + bpo + octaves)
    rng = np.random.default_rng(3)
    for f in range(x.shape[0]):
        ts = int(rng.integers(8_000_000, 30_000_000)) if jitter else 16_000_000
        a.preprocess(x[f], ts); b.preprocess(x[f], ts)
        assert same_bits([e.y for e in a.smoothed], b.sm), f
        assert np.array_equal(a.peaks, b.peaks) and same_bits(a.centers, b.centers) and same_bits(a.sizes, b.sizes), f
        assert same_bits(a.peakfiltered, b.peakfiltered) and same_bits(a.afterglow, b.afterglow), f
        assert same_bits([e.y for e in a.calm], b.calm) and same_bits([e.y for e in a.released], b.released), f
        assert same_bits(a.pitch_accuracy, b.pitch_accuracy) and same_bits(a.pitch_deviation, b.pitch_deviation), f
        assert same_bits(a.scene.y, b.scene) and same_bits(a.tuning.y, b.tuning), f
    assert b.scene > 0 and np.abs(b.calm).max() > 0     # the recurrences moved
