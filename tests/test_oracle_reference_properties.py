"""The reference's own tests, ported 1:1 onto the CPU oracle (same stimuli, same thresholds).
These are what pin the oracle: the reference ships no golden vectors."""
import numpy as np

import oracle as O


def _sweep_max_and_sum(ov, p, freqs):
    mx, sm = [], []
    for f in freqs:
        db = ov.calculate_vqt_instant_in_db(O.test_create_sines(p, [f]))
        mx.append(db.max())
        sm.append(db.sum(dtype=np.float32))
    return np.array(mx), np.array(sm)


def test_vqt_bandwidths():
    """vqt.rs:996-1027: no coverage holes across 588x20 log-spaced sines.  Every 4th point here keeps the CPU suite
    short; tests/test_reference_properties_gpu.py::test_vqt_bandwidths_full_sweep runs all 11 740 points through the
    HIP path (both algorithms) and through this oracle on the GPU box."""
    p = O.default_params()
    ov = O.OracleVqt(p)
    sub = 20
    n = p.n_buckets()
    idx = np.arange(sub // 2, n * sub - sub // 2)[::4]
    freqs = np.float32(p.min_freq) * np.power(np.float32(2.0), idx.astype(np.float32) / np.float32(p.buckets_per_octave * sub))
    mx, sm = _sweep_max_and_sum(ov, p, freqs)
    assert mx.max() - sm.min() < 3.0


def test_vqt_group_boundary_continuity():
    """vqt.rs:1032-1076"""
    p = O.default_params()
    ov = O.OracleVqt(p)
    freq, _, M, _ = ov.filter_params()
    boundaries = [freq[i + 1] for i in range(len(M) - 1) if M[i] != M[i + 1]]
    assert boundaries
    steps = 20
    for b in boundaries:
        fs = [np.float32(b) * np.float32(2.0) ** np.float32(i / (steps * 4.0 * 12.0)) for i in range(-steps, steps + 1)]
        mx, _ = _sweep_max_and_sum(ov, p, fs)
        assert mx.max() - mx.min() < 3.0, f"spread {mx.max() - mx.min():.2f} dB at {b:.1f} Hz"


def test_vqt_delay():
    """vqt.rs:1078-1085 (98 ms per VQT_REVIEW.md:363)"""
    ov = O.OracleVqt(O.default_params())
    assert int(ov.delay * 1000) < 100
    assert int(ov.delay * 1000) == 98


def test_fft_library():
    """vqt.rs:1087-1103: forward+inverse complex FFT is unnormalised"""
    x = np.zeros(256, np.complex64)
    x[0] = 1.0
    y = O.fft_complex(O.fft_complex(x), inverse=True)
    assert y[0].real == 256.0


def test_real_fft_library():
    """vqt.rs:1105-1128: R2C half spectrum == lower half of the complex FFT"""
    n = 256
    sig = np.sin(np.arange(n, dtype=np.float32) * np.float32(0.1)).astype(np.float32)
    full = O.fft_complex(sig.astype(np.complex64))
    half = O.fft_real(sig)
    assert half.size == n // 2 + 1
    assert np.abs(half - full[: n // 2 + 1]).max() < 1e-3
    # and both agree with numpy (sign / layout convention)
    assert np.abs(full - np.fft.fft(sig.astype(np.float64))).max() < 1e-3


def test_vqt_close_frequencies():
    """lib.rs:16-48: two sines a semitone apart => exactly 2 peaks (fresh AnalysisState with a
    1100 ms step: EMA alpha ~ 1, i.e. the stateless peak pipeline on 0.99999..*frame)."""
    p = O.default_params()
    ov = O.OracleVqt(p)
    sub = 30
    counts = []
    for i in range(int(2.6 * sub), p.octaves * sub - sub // 2):
        ln = np.float32(i) / np.float32(sub)
        f1 = np.float32(p.min_freq) * np.float32(2.0) ** ln
        f2 = np.float32(p.min_freq) * np.float32(2.0) ** (ln + np.float32(1.0 / 12.0))
        db = ov.calculate_vqt_instant_in_db(O.test_create_sines(p, [f1, f2]))
        # EmaMeasurement::update_with_timestep from y=0 (util.rs:106-125): y = alpha * x
        horizon_ms = (70.0 * (1.5 - 0.5 * (np.arange(p.n_buckets(), dtype=np.float32) / p.buckets_per_octave / p.octaves)) * 0.6).astype(np.int64)
        alpha = (1.0 - np.exp(-2.0 * 1.1 / (horizon_ms / 1000.0))).astype(np.float32)
        sm = (alpha * db).astype(np.float32)
        counts.append(len(O.find_peaks_split(sm, p.buckets_per_octave)))
    assert all(c == 2 for c in counts), counts


def test_vqt_high_frequencies():
    """lib.rs:50-72"""
    p = O.default_params()
    ov = O.OracleVqt(p)
    sub = 30
    fs = [np.float32(p.min_freq) * np.float32(2.0) ** (np.float32(i) + np.float32(j) / np.float32(12.0 * sub))
          for i in range(p.octaves) for j in range(sub)]
    mx, _ = _sweep_max_and_sum(ov, p, fs)
    assert mx.min() > mx.max() - 6.0


def test_analysis_does_something():
    """analysis.rs:415-428: a zero frame yields no peaks / zero smoothed values"""
    z = np.zeros(48, np.float32)
    idx, ce, sz = O.analyze_frame(z, 55.0, 2, 24)
    assert idx.size == 0


def test_n_buckets_doctest():
    """vqt.rs:224-237"""
    assert O.OracleParams(min_freq=55.0, octaves=7, buckets_per_octave=84).n_buckets() == 7 * 84
