"""Filter::bandwidth_3db_in_hz and the kernel construction's coverage-gap warnings (vqt.rs:421, :695-709, :817-818, :956-989)
as the product's host code reports them (no GPU: a handle without a device answers getters).  Checked against an independent
NumPy float64 restatement of the un-sparsified decimated frequency response, and through the properties the reference's own
log lines state."""
import re

import numpy as np
import pytest

import pitchvis_amd as P


def _numpy_bands(v):
    """calculate_filter's window -> FFT -> |.| (vqt.rs:793-815) in float64, then find_3db_points / calculate_bandwidth."""
    p = v.params()
    freq, wl, m, _ = v.filter_params()
    ker = v.kernel()
    n_fft = p.n_fft
    center = n_fft - float(wl[0]) / 2.0
    lo = np.zeros(v.n_bins); hi = np.zeros(v.n_bins)
    k = 0
    for g in ker.window_groups:
        w0, w1 = g.window
        for _ in range(g.filter_bank.shape[0]):
            s = int(m[k]); nf = (w1 - w0) // s
            length = int(np.round(np.float32(wl[k]) / np.float32(s)))
            c = int(np.floor((np.float32(center) - np.float32(w0)) / np.float32(s)))
            beg = c - length // 2
            i = np.arange(length)
            x = np.zeros(nf, complex)
            x[beg:beg + length] = (0.5 - 0.5 * np.cos(2 * np.pi * i / (length - 1))) * np.exp(2j * np.pi * i * float(freq[k]) * s / p.sr)
            mag = np.abs(np.fft.fft(x / np.abs(x).sum()))
            pk = int(np.argmax(mag)); thr = mag[pk] / np.sqrt(2.0)
            a = pk
            while a > 0 and mag[a] > thr:
                a -= 1
            b = pk
            while b < nf - 1 and mag[b] > thr:
                b += 1
            lo[k] = a * (p.sr / s) / nf; hi[k] = b * (p.sr / s) / nf
            k += 1
    return lo, hi


@pytest.mark.parametrize("kw", [dict(), dict(sr=48000.0, range=P.VqtRange(55.0, 7, 36))])
def test_bandwidths_match_a_float64_restatement(kw):
    v = P.Vqt(P.VqtParameters(**kw), device=None)
    lo, hi = v.bandwidth_3db_in_hz
    wlo, whi = _numpy_bands(v)
    freq, _, m, _ = v.filter_params()
    # the f32 response and the f64 one can disagree about a sample that sits on the -3 dB threshold: at most one bucket, rarely
    ker = v.kernel()
    bucket = np.concatenate([np.full(g.filter_bank.shape[0], 1.0) for g in ker.window_groups])
    k = 0
    for g in ker.window_groups:
        n = g.filter_bank.shape[0]
        bucket[k:k + n] = (v.params().sr / m[k:k + n]) / ((g.window[1] - g.window[0]) // m[k:k + n])
        k += n
    assert (np.abs(lo - wlo) <= bucket * 1.001).all() and (np.abs(hi - whi) <= bucket * 1.001).all()
    assert np.mean(lo == wlo.astype(np.float32)) > 0.97 and np.mean(hi == whi.astype(np.float32)) > 0.97
    # the centre frequency lies inside its own band, and the band is a few buckets wide ("a very crude approximation", vqt.rs:960)
    assert (lo <= freq).all() and (freq <= hi).all()
    assert ((hi - lo) / bucket >= 1.999).all()


def test_default_kernel_has_no_coverage_gap_and_a_sharper_one_warns_like_the_reference():
    v = P.Vqt(P.VqtParameters(), device=None)
    assert v.warnings == []   # the reference's own record: "the now-alive gap check confirms full -3 dB coverage (no warnings)", VQT_REVIEW.md:371-372
    q = 4.0
    w = P.Vqt(P.VqtParameters(quality=q, gamma=4.8 * q), device=None)
    lo, hi = w.bandwidth_3db_in_hz
    freq = w.filter_params()[0]
    gaps = [k for k in range(1, w.n_bins) if lo[k] > hi[k - 1] and hi[k - 1] > 0.0]
    assert len(w.warnings) == len(gaps) > 0
    pat = re.compile(r"coverage gap below the filter at ([0-9.]+) Hz: its -3 dB band starts at ([0-9.]+) Hz but the previous filter's "
                     r"band ends at ([0-9.]+) Hz \(([0-9.]+)% of this filter's bandwidth\); decrease quality to close the gap$")
    for k, line in zip(gaps, w.warnings):
        mt = pat.match(line)
        assert mt, line
        f, a, b, pct = (float(x) for x in mt.groups())
        assert abs(f - freq[k]) <= 0.051 and abs(a - lo[k]) <= 0.0051 and abs(b - hi[k - 1]) <= 0.0051
        assert abs(pct - 100.0 * (lo[k] - hi[k - 1]) / (hi[k] - lo[k])) <= 0.051


def test_null_and_range_arguments():
    import ctypes as C
    v = P.Vqt(P.VqtParameters(), device=None)
    L = v._L
    assert L.pvq_vqt_bandwidths_3db(v._h, None, None) != 0
    buf = C.create_string_buffer(8)
    assert L.pvq_vqt_warning(v._h, 0, buf, 8) != 0          # no warnings: index out of range
    assert L.pvq_vqt_warning_count(None) == 0
