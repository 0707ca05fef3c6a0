"""ctypes loader for the CPU ORACLE (oracle/pvq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pitchvis_amd/ may import this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PVQ_ORACLE_LIB: another build of the same source (bench.py's cpu_baseline times oracle/liboracle_native.so, -O3 -march=native, as
# a courtesy figure; tests/test_sanitize_cpu.py loads the ASan / UBSan build)
_LIB_PATH = os.environ.get("PVQ_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (idempotent)."""
    src = os.path.join(_HERE, "pvq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [
        ("sr", C.c_float),
        ("n_fft", C.c_uint32),
        ("min_freq", C.c_float),
        ("octaves", C.c_uint32),
        ("buckets_per_octave", C.c_uint32),
        ("sparsity_quantile", C.c_float),
        ("quality", C.c_float),
        ("gamma", C.c_float),
    ]


class _AParams(C.Structure):
    _fields_ = [
        ("peak_min_prominence", C.c_float),
        ("peak_min_height", C.c_float),
        ("bass_min_prominence", C.c_float),
        ("bass_min_height", C.c_float),
        ("highest_bassnote", C.c_uint32),
        ("harmonic_threshold", C.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        up = C.POINTER(C.c_uint32)
        vp = C.c_void_p
        L.orc_default_params.argtypes = [C.POINTER(_Params)]
        L.orc_vqt_new.argtypes = [C.POINTER(_Params), C.POINTER(vp), fp]
        L.orc_vqt_new.restype = C.c_int
        L.orc_vqt_free.argtypes = [vp]
        L.orc_n_bins.argtypes = [vp]; L.orc_n_bins.restype = C.c_uint32
        L.orc_n_groups.argtypes = [vp]; L.orc_n_groups.restype = C.c_uint32
        L.orc_delay_seconds.argtypes = [vp]; L.orc_delay_seconds.restype = C.c_double
        L.orc_window_center.argtypes = [vp]; L.orc_window_center.restype = C.c_float
        L.orc_filter_params.argtypes = [vp, fp, fp, up, up]
        L.orc_group_info.argtypes = [vp, C.c_uint32, up]
        L.orc_group_csr.argtypes = [vp, C.c_uint32, C.c_int, up, up, fp]
        L.orc_calculate_vqt_instant_in_db.argtypes = [vp, fp, fp]
        L.orc_calculate_vqt_instant_complex.argtypes = [vp, fp, fp]
        L.orc_group_spectrum.argtypes = [vp, C.c_uint32, fp, fp]
        L.orc_power_to_db.argtypes = [fp, C.c_uint32, fp]
        L.orc_calculate_batch.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_size_t, fp, fp]
        L.orc_test_create_sines.argtypes = [C.POINTER(_Params), fp, C.c_uint32, C.c_float, fp]
        L.orc_find_peaks.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, up]
        L.orc_find_peaks.restype = C.c_uint32
        L.orc_default_analysis_params.argtypes = [C.POINTER(_AParams)]
        L.orc_find_peaks_split.argtypes = [fp, C.c_uint32, C.c_uint32, C.POINTER(_AParams), up]
        L.orc_find_peaks_split.restype = C.c_uint32
        L.orc_enhance_peaks_continuous.argtypes = [up, C.c_uint32, fp, C.c_float, C.c_uint32, C.c_uint32, fp, fp]
        L.orc_enhance_peaks_continuous.restype = C.c_uint32
        L.orc_promote_bass_peaks_with_harmonics.argtypes = [fp, fp, C.c_uint32, fp, C.c_float, C.c_uint32,
                                                            C.c_uint32, C.c_uint32, C.c_float]
        L.orc_analyze_frame.argtypes = [fp, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32,
                                        C.POINTER(_AParams), up, fp, fp]
        L.orc_analyze_frame.restype = C.c_uint32
        L.orc_expf.argtypes = [C.c_float]; L.orc_expf.restype = C.c_float
        L.orc_powf.argtypes = [C.c_float, C.c_float]; L.orc_powf.restype = C.c_float
        L.orc_expf_v.argtypes = [fp, C.c_uint32, fp]
        L.orc_powf_v.argtypes = [C.c_float, fp, C.c_uint32, fp]
        L.orc_fft_complex.argtypes = [fp, C.c_uint32, C.c_int]
        L.orc_fft_real.argtypes = [fp, C.c_uint32, fp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


@dataclass
class OracleParams:
    """VqtParameters + VqtRange flattened (vqt.rs:238-262, 278-348)."""
    sr: float = 22050.0
    n_fft: int = 32768
    min_freq: float = 55.0
    octaves: int = 7
    buckets_per_octave: int = 84
    sparsity_quantile: float = 0.999
    quality: float = 1.6
    gamma: float = 4.8 * 1.6

    def n_buckets(self) -> int:
        return self.octaves * self.buckets_per_octave

    def c(self) -> _Params:
        return _Params(self.sr, self.n_fft, self.min_freq, self.octaves, self.buckets_per_octave,
                       self.sparsity_quantile, self.quality, self.gamma)


def default_params() -> OracleParams:
    p = _Params()
    lib().orc_default_params(C.byref(p))
    return OracleParams(p.sr, p.n_fft, p.min_freq, p.octaves, p.buckets_per_octave,
                        p.sparsity_quantile, p.quality, p.gamma)


@dataclass
class OracleAnalysisParams:
    """peak-related part of AnalysisParameters::default (analysis.rs:72-98)."""
    peak_min_prominence: float = 10.0
    peak_min_height: float = 4.0
    bass_min_prominence: float = 5.0
    bass_min_height: float = 3.5
    highest_bassnote: int = 28
    harmonic_threshold: float = 0.3

    def c(self) -> _AParams:
        return _AParams(self.peak_min_prominence, self.peak_min_height, self.bass_min_prominence,
                        self.bass_min_height, self.highest_bassnote, self.harmonic_threshold)


class OracleVqtError(Exception):
    def __init__(self, code, a, b):
        self.code, self.a, self.b = code, a, b
        name = {1: "AboveNyquist", 2: "WindowExceedsNFft"}[code]
        super().__init__(f"{name}({a}, {b})")


class OracleVqt:
    """CPU restatement of pitchvis_analysis::vqt::Vqt."""

    def __init__(self, params: OracleParams):
        self.params = params
        self._h = C.c_void_p()
        err = (C.c_float * 2)()
        cp = params.c()
        rc = lib().orc_vqt_new(C.byref(cp), C.byref(self._h), err)
        if rc != 0:
            raise OracleVqtError(rc, err[0], err[1])
        self.n_bins = lib().orc_n_bins(self._h)
        self.n_groups = lib().orc_n_groups(self._h)
        self.delay = lib().orc_delay_seconds(self._h)
        self.window_center = lib().orc_window_center(self._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().orc_vqt_free(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def filter_params(self):
        n = self.n_bins
        freq = np.empty(n, np.float32); wl = np.empty(n, np.float32)
        m = np.empty(n, np.uint32); mw = np.empty(n, np.uint32)
        lib().orc_filter_params(self._h, _f(freq), _f(wl), _u(m), _u(mw))
        return freq, wl, m, mw

    def group_info(self, g):
        info = np.zeros(5, np.uint32)
        lib().orc_group_info(self._h, g, _u(info))
        return dict(window=(int(info[0]), int(info[1])), rows=int(info[2]), nnz=int(info[3]), neg_nnz=int(info[4]))

    def group_csr(self, g, neg=False):
        gi = self.group_info(g)
        nnz = gi["neg_nnz"] if neg else gi["nnz"]
        rp = np.zeros(gi["rows"] + 1, np.uint32)
        ci = np.zeros(max(nnz, 1), np.uint32)
        va = np.zeros(2 * max(nnz, 1), np.float32)
        lib().orc_group_csr(self._h, g, 1 if neg else 0, _u(rp), _u(ci), _f(va))
        return rp, ci[:nnz], va[: 2 * nnz].view(np.complex64)

    def calculate_vqt_instant_in_db(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.params.n_fft,), "input must be exactly n_fft samples"
        out = np.empty(self.n_bins, np.float32)
        lib().orc_calculate_vqt_instant_in_db(self._h, _f(x), _f(out))
        return out

    def calculate_vqt_instant_complex(self, x):
        x = np.ascontiguousarray(x, np.float32)
        assert x.shape == (self.params.n_fft,)
        out = np.empty(2 * self.n_bins, np.float32)
        lib().orc_calculate_vqt_instant_complex(self._h, _f(x), _f(out))
        return out.view(np.complex64)

    def group_spectrum(self, g, x):
        x = np.ascontiguousarray(x, np.float32)
        gi = self.group_info(g)
        n = (gi["window"][1] - gi["window"][0]) // 2 + 1
        out = np.empty(2 * n, np.float32)
        lib().orc_group_spectrum(self._h, g, _f(x), _f(out))
        return out.view(np.complex64)

    def calculate_batch(self, pcm, hop, n_frames, n_lead=0, want_complex=False):
        pcm = np.ascontiguousarray(pcm, np.float32)
        assert pcm.size >= n_lead + n_frames * hop
        out = np.empty((n_frames, self.n_bins), np.float32)
        oc = np.empty((n_frames, self.n_bins), np.complex64) if want_complex else None
        lib().orc_calculate_batch(self._h, _f(pcm), n_lead, hop, n_frames, _f(out),
                                  oc.ctypes.data_as(C.POINTER(C.c_float)) if want_complex else None)
        return (out, oc) if want_complex else out


def power_to_db(xc):
    xc = np.ascontiguousarray(xc, np.complex64)
    out = np.empty(xc.size, np.float32)
    lib().orc_power_to_db(xc.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), xc.size, _f(out))
    return out


def test_create_sines(params: OracleParams, freqs, t_diff=0.0):
    fr = np.asarray(freqs, np.float32)
    wave = np.empty(params.n_fft, np.float32)
    cp = params.c()
    lib().orc_test_create_sines(C.byref(cp), _f(fr), fr.size, t_diff, _f(wave))
    return wave


test_create_sines.__test__ = False  # not a pytest test


def find_peaks(vqt, buckets_per_octave, min_prominence, min_height):
    vqt = np.ascontiguousarray(vqt, np.float32)
    out = np.empty(max(vqt.size, 1), np.uint32)
    n = lib().orc_find_peaks(_f(vqt), vqt.size, buckets_per_octave, min_prominence, min_height, _u(out))
    return out[:n].copy()


def find_peaks_split(vqt, buckets_per_octave, ap: OracleAnalysisParams | None = None):
    ap = ap or OracleAnalysisParams()
    vqt = np.ascontiguousarray(vqt, np.float32)
    out = np.empty(max(vqt.size, 1), np.uint32)
    cap = ap.c()
    n = lib().orc_find_peaks_split(_f(vqt), vqt.size, buckets_per_octave, C.byref(cap), _u(out))
    return out[:n].copy()


def analyze_frame(vqt, min_freq, octaves, buckets_per_octave, ap: OracleAnalysisParams | None = None):
    """analysis.rs:332-361 with pass-through smoothing -> (peaks, centers, sizes)."""
    ap = ap or OracleAnalysisParams()
    vqt = np.ascontiguousarray(vqt, np.float32)
    idx = np.empty(max(vqt.size, 1), np.uint32)
    ce = np.empty(max(vqt.size, 1), np.float32)
    sz = np.empty(max(vqt.size, 1), np.float32)
    cap = ap.c()
    n = lib().orc_analyze_frame(_f(vqt), vqt.size, min_freq, octaves, buckets_per_octave, C.byref(cap),
                                _u(idx), _f(ce), _f(sz))
    return idx[:n].copy(), ce[:n].copy(), sz[:n].copy()


def enhance_peaks_continuous(peaks, vqt, min_freq, octaves, buckets_per_octave):
    peaks = np.ascontiguousarray(peaks, np.uint32)
    vqt = np.ascontiguousarray(vqt, np.float32)
    ce = np.empty(max(peaks.size, 1), np.float32)
    sz = np.empty(max(peaks.size, 1), np.float32)
    lib().orc_enhance_peaks_continuous(_u(peaks), peaks.size, _f(vqt), min_freq, octaves, buckets_per_octave,
                                       _f(ce), _f(sz))
    return ce[: peaks.size].copy(), sz[: peaks.size].copy()


def promote_bass_peaks_with_harmonics(center, size, vqt, min_freq, octaves, buckets_per_octave,
                                      highest_bassnote=28, harmonic_threshold=0.3):
    center = np.ascontiguousarray(center, np.float32)
    size = np.array(size, np.float32, copy=True)
    vqt = np.ascontiguousarray(vqt, np.float32)
    lib().orc_promote_bass_peaks_with_harmonics(_f(center), _f(size), center.size, _f(vqt), min_freq, octaves,
                                                buckets_per_octave, highest_bassnote, harmonic_threshold)
    return size


def fft_complex(a, inverse=False):
    a = np.array(a, np.complex64, copy=True)
    lib().orc_fft_complex(a.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), a.size, 1 if inverse else 0)
    return a


def fft_real(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size // 2 + 1, np.complex64)
    lib().orc_fft_real(_f(x), x.size, out.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)))
    return out


def expf(x) -> np.float32:
    return np.float32(lib().orc_expf(float(np.float32(x))))


def powf(x, y) -> np.float32:
    return np.float32(lib().orc_powf(float(np.float32(x)), float(np.float32(y))))


def expf_v(x) -> np.ndarray:
    """glibc expf over an f32 array (the same call as expf, per element)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_expf_v(_f(x), x.size, _f(out))
    return out


def powf_v(base, y) -> np.ndarray:
    y = np.ascontiguousarray(y, np.float32)
    out = np.empty_like(y)
    lib().orc_powf_v(float(np.float32(base)), _f(y), y.size, _f(out))
    return out
