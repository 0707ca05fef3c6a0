"""Host-side AnalysisState: the product's C++ restatement against the oracle's
NumPy-f32 restatement, the reference's own tests for this layer, and the API contract."""
import numpy as np
import pytest

import oracle as O
from oracle.analysis_state import Ema, OracleAnalysisState
import pitchvis_amd as P
from helpers import white_noise
from synth import piano_roll

MS = 1_000_000


def test_ema_basic_and_limit():
    """util.rs:143-225: frame-rate independence of the EMA"""
    lo, hi = Ema(1000 * MS, 0.0), Ema(1000 * MS, 0.0)
    for v in (1.0, 2.0, 3.0, 4.0):
        for _ in range(2):
            lo.update(v, 250 * MS)
        for _ in range(4):
            hi.update(v, 125 * MS)
    assert abs(lo.y - hi.y) < 0.05
    e_hi, e_med, e_lo = Ema(1000 * MS, 0.0), Ema(1000 * MS, 0.0), Ema(1000 * MS, 0.0)
    for _ in range(100):
        e_hi.update(1.0, (500 // 100) * MS)
    for _ in range(10):
        e_med.update(1.0, (500 // 10) * MS)
    for _ in range(3):
        e_lo.update(1.0, (500 // 3) * MS)
    assert abs(e_lo.y - e_hi.y) < 0.02 and abs(e_lo.y - e_med.y) < 0.02
    assert abs(e_lo.y - (1 - np.exp(-1))) < 0.02


def test_analysis_does_something():
    """analysis.rs:415-428"""
    st = P.AnalysisState.new(P.VqtRange(55.0, 2, 24))
    st.preprocess(np.zeros(48, np.float32), 1.0)
    assert st.x_vqt_smoothed.size == 48 and (st.x_vqt_smoothed == 0).all()
    with pytest.raises(AssertionError):  # analysis.rs:289
        st.preprocess(np.zeros(47, np.float32), 1.0)
    # doc-test analysis.rs:110-118
    st2 = P.AnalysisState.new(P.VqtRange(55.0, 8, 24), P.FullAnalysisParameters())
    st2.preprocess(np.zeros(8 * 24, np.float32), 0.030)
    assert abs(st.bin_to_frequency(24) - 110.0) < 1e-3


def _frames(seed=3, seconds=4.0, hop=2048):
    op = O.OracleParams(sr=48000.0, octaves=7, buckets_per_octave=36)
    ov = O.OracleVqt(op)
    pcm, _ = piano_roll(op.sr, seconds, seed)
    pcm = pcm + white_noise(pcm.size, seed, amp=0.01)
    return op, ov.calculate_batch(pcm.astype(np.float32), hop, pcm.size // hop), hop / op.sr


@pytest.mark.parametrize("mode", ["default", "no_smoothing", "retuned"])
def test_product_matches_oracle_frame_by_frame(mode):
    op, frames, dt = _frames()
    st = P.AnalysisState.new(P.VqtRange(op.min_freq, op.octaves, op.buckets_per_octave))
    ost = OracleAnalysisState(op.min_freq, op.octaves, op.buckets_per_octave)
    ts_ns = int(round(dt * 1e9))
    if mode == "no_smoothing":   # viewer's VQTSmoothingMode::None
        st.update_vqt_smoothing_duration(None)
        ost.update_vqt_smoothing_duration(None)
    elif mode == "retuned":
        st.update_vqt_smoothing_duration(0.120)
        ost.update_vqt_smoothing_duration(120 * MS)
    for f in range(frames.shape[0]):
        st.preprocess(frames[f], ts_ns / 1e9)
        ost.preprocess(frames[f], ts_ns)
        assert sorted(st.peaks) == list(ost.peaks), (mode, f)
        assert np.allclose(st.x_vqt_smoothed, [e.y for e in ost.smoothed], rtol=2e-6, atol=1e-6)
        assert np.allclose(st.x_vqt_peakfiltered, ost.peakfiltered, rtol=2e-6, atol=1e-6)
        assert np.allclose(st.x_vqt_afterglow, ost.afterglow, rtol=2e-6, atol=1e-6)
        assert np.allclose(st.calmness, [e.y for e in ost.calm], rtol=1e-5, atol=1e-6)
        pc = st.peaks_continuous
        assert len(pc) == ost.centers.size
        # same input, tight: the oracle's peak pipeline on the product's own smoothed frame
        _, wce, wsz = O.analyze_frame(st.x_vqt_smoothed, op.min_freq, op.octaves, op.buckets_per_octave)
        assert np.allclose([p.center for p in pc], wce, atol=1e-5) and np.allclose([p.size for p in pc], wsz, atol=1e-5)
        # whole chain, loose: the f32 log-frequency parabola (peak_detection.rs:91-118) amplifies the 1-ulp
        # differences between the two EMA evaluations to ~1e-2 bins
        assert np.allclose([p.center for p in pc], ost.centers, atol=3e-2) and np.allclose([p.size for p in pc], ost.sizes, atol=0.3)
        assert np.allclose(st.pitch_accuracy, ost.pitch_accuracy, atol=2e-2)
        assert np.allclose(st.pitch_deviation, ost.pitch_deviation, atol=1e-2)
        assert abs(st.smoothed_scene_calmness - ost.scene.y) < 1e-5
        assert abs(st.smoothed_tuning_grid_inaccuracy - ost.tuning.y) < 0.2   # cents; same amplification
    if mode == "no_smoothing":
        assert np.array_equal(st.x_vqt_smoothed, frames[-1])
    assert st.smoothed_scene_calmness > 0.0  # the recurrence actually moved


def test_vqt_close_frequencies_with_the_real_state():
    """lib.rs:16-48 with the product's AnalysisState on oracle dB frames: exactly 2 peaks"""
    p = O.default_params()
    ov = O.OracleVqt(p)
    sub = 30
    for i in range(int(2.6 * sub), p.octaves * sub - sub // 2, 7):
        ln = np.float32(i) / np.float32(sub)
        f1 = np.float32(p.min_freq) * np.float32(2.0) ** ln
        f2 = np.float32(p.min_freq) * np.float32(2.0) ** (ln + np.float32(1.0 / 12.0))
        db = ov.calculate_vqt_instant_in_db(O.test_create_sines(p, [f1, f2]))
        st = P.AnalysisState.new(P.VqtRange(p.min_freq, p.octaves, p.buckets_per_octave))
        st.preprocess(db, 1.100)
        assert len(st.peaks) == 2, i


@pytest.mark.gpu
def test_gpu_frames_drive_the_host_state():
    """GPU dB frames -> host AnalysisState equals the all-CPU chain up to the dB parity tolerance."""
    torch = pytest.importorskip("torch")
    op, wframes, dt = _frames(seed=9, seconds=3.0)
    pp = P.VqtParameters(sr=op.sr, range=P.VqtRange(op.min_freq, op.octaves, op.buckets_per_octave))
    v = P.Vqt.new(pp, 0)
    pcm, _ = piano_roll(op.sr, 3.0, 9)
    pcm = (pcm + white_noise(pcm.size, 9, amp=0.01)).astype(np.float32)
    gframes = v.calculate_batch_db(pcm, 2048, pcm.size // 2048)
    assert np.abs(gframes - wframes).max() <= 1e-2
    a, b = P.AnalysisState.new(pp.range), P.AnalysisState.new(pp.range)
    for f in range(gframes.shape[0]):
        a.preprocess(gframes[f], dt)
        b.preprocess(wframes[f], dt)
    assert np.abs(a.x_vqt_smoothed - b.x_vqt_smoothed).max() <= 1e-2
    assert abs(a.smoothed_scene_calmness - b.smoothed_scene_calmness) < 1e-3
