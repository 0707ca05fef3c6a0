"""find_peaks 0.1.5 semantics as restated in the oracle, cross-checked against scipy.signal."""
import numpy as np
import pytest
from scipy.signal import find_peaks as sp_find_peaks

import oracle as O


def _scipy(x, bpo, prom, height):
    dist = int(np.floor(bpo * 0.4 / 12.0 + 0.5))
    kw = dict(height=height, prominence=prom)
    if dist >= 1:
        kw["distance"] = dist
    pk, _ = sp_find_peaks(x.astype(np.float64), **kw)
    min_bin = ((bpo // 12) + 1) // 2
    return pk[pk >= min_bin]


@pytest.mark.parametrize("bpo", [12, 36, 84])
@pytest.mark.parametrize("seed", range(6))
def test_random_frames_match_scipy(bpo, seed):
    """no plateaus in random data -> scipy.signal.find_peaks is an exact stand-in"""
    rng = np.random.default_rng(seed)
    n = bpo * 7
    x = np.abs(rng.normal(0, 8, n)).astype(np.float32)
    x = np.convolve(x, np.ones(3) / 3, mode="same").astype(np.float32)
    for prom, height in ((10.0, 4.0), (5.0, 3.5), (1.0, 0.0)):
        got = O.find_peaks(x, bpo, prom, height)
        want = _scipy(x, bpo, prom, height)
        assert np.array_equal(got, want), (bpo, seed, prom, got, want)


def test_edges_are_never_peaks_and_min_bin():
    x = np.array([9, 1, 1, 8, 1, 1, 1, 9], np.float32)
    assert list(O.find_peaks(x, 12, 1.0, 0.0)) == [3]
    x = np.array([0, 9, 0, 0, 9, 0], np.float32)  # bpo 84 -> min_bin 4
    assert list(O.find_peaks(x, 84, 1.0, 0.0)) == [4]
    assert list(O.find_peaks(x, 36, 1.0, 0.0)) == [4]  # bpo 36 -> min_bin 2
    assert list(O.find_peaks(x, 12, 1.0, 0.0)) == [1, 4]


def test_plateau_middle_position():
    # Peak.position = start..end (half-open); middle_position = (start+end)/2
    x = np.array([0, 1, 5, 5, 5, 1, 0, 7, 7, 0], np.float32)
    assert list(O.find_peaks(x, 12, 0.5, 0.0)) == [3, 8]   # [2,5)->3 ; [7,9)->8
    x = np.array([0, 5, 5, 6, 0], np.float32)             # rising plateau is not a peak, 6 is
    assert list(O.find_peaks(x, 12, 0.5, 0.0)) == [3]


def test_prominence_and_inclusive_bounds():
    x = np.array([0, 10, 6, 8, 0, 0], np.float32)
    assert list(O.find_peaks(x, 12, 2.0, 0.0)) == [1, 3]    # prominence of 8 is exactly 2 (inclusive)
    assert list(O.find_peaks(x, 12, 2.0001, 0.0)) == [1]
    assert list(O.find_peaks(x, 12, 0.0, 10.0)) == [1]      # height inclusive
    assert list(O.find_peaks(x, 12, 0.0, 10.0001)) == []


def test_bass_general_split():
    """analysis.rs:332-349: bass cfg (prom 5, height 3.5) for p <= 28, general (10, 4) above"""
    x = np.zeros(252, np.float32)
    x[10] = 6.0     # bass: prom 6 >= 5 -> kept
    x[28] = 6.0     # p == 28 is still bass
    x[29 + 1] = 6.0  # general: prom 6 < 10 -> dropped
    x[100] = 12.0   # general kept
    x[150] = 3.9    # below general height
    assert list(O.find_peaks_split(x, 36)) == [10, 28, 100]


def test_enhance_and_promote_known_cases():
    n, bpo, oct_ = 252, 36, 7
    x = np.zeros(n, np.float32)
    x[99:102] = [10.0, 20.0, 10.0]   # symmetric in ln f -> centre stays on the bin
    x[0] = 5.0                        # edge case: passthrough
    ce, sz = O.enhance_peaks_continuous(np.array([100, 0], np.uint32), x, 55.0, oct_, bpo)
    # the f32 Lagrange fit in ln-frequency is ill-conditioned (abscissae ~6.6 that differ by 0.019):
    # the reference formula itself lands ~6e-3 bins off centre on exactly symmetric input
    assert list(ce) == sorted(ce) and abs(ce[1] - 100.0) < 2e-2 and abs(sz[1] - 20.0) < 0.25
    assert ce[0] == 0.0 and sz[0] == 5.0
    # promotion: fundamental at bin 12 (A1 + 4 semitones) with a strong 2f (bin 48) -> +<=1.76 dB
    y = np.zeros(n, np.float32)
    y[12] = 20.0
    y[48] = 20.0
    s2 = O.promote_bass_peaks_with_harmonics(np.array([12.0], np.float32), np.array([20.0], np.float32), y, 55.0, oct_, bpo)
    assert abs(s2[0] - (20.0 + 10 * np.log10(1.25))) < 1e-3
    y[48] = 0.0
    s3 = O.promote_bass_peaks_with_harmonics(np.array([12.0], np.float32), np.array([20.0], np.float32), y, 55.0, oct_, bpo)
    assert s3[0] == 20.0
    # not a bass note -> untouched
    s4 = O.promote_bass_peaks_with_harmonics(np.array([40.0], np.float32), np.array([20.0], np.float32), y, 55.0, oct_, bpo)
    assert s4[0] == 20.0
