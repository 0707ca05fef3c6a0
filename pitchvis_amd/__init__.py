"""pitchvis_amd — MI355X-native batched VQT pitch-analysis engine (Python face of libpvq).

Mirrors the public surface of the reference crate `pitchvis_analysis` for its hot path
(reference files pitchvis_analysis/src/vqt.rs, analysis.rs, analysis_modules/peak_detection.rs):

    VqtRange, VqtParameters, VqtError{AboveNyquist, WindowExceedsNFft}, WindowGroup, VqtKernel,
    Vqt.new / params() / kernel() / delay / calculate_vqt_instant_in_db,
    PeakDetectionParameters, AnalysisParameters, ContinuousPeak

plus the batched forms the GPU makes worthwhile.  Everything computes on the GPU through the
C ABI in include/pvq.h; there is no CPU fallback (a handle without a device refuses to compute).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _lib
from ._lib import ALGO_AUTO, ALGO_BLOCKDFT, ALGO_FFT, GEMM_BF16X3, GEMM_F32  # noqa: F401

__all__ = [
    "VqtRange", "VqtParameters", "VqtError", "AboveNyquist", "WindowExceedsNFft", "PvqError", "WindowGroup",
    "VqtKernel", "Vqt", "PeakDetectionParameters", "AnalysisParameters", "ContinuousPeak", "FrameAnalysis",
    "ALGO_AUTO", "ALGO_FFT", "ALGO_BLOCKDFT", "GEMM_F32", "GEMM_BF16X3", "AnalysisState", "AnalysisBatch", "FullAnalysisParameters",
]


# ---- vqt.rs:238-262 -------------------------------------------------------------------------
@dataclass
class VqtRange:
    min_freq: float = 55.0
    octaves: int = 7
    buckets_per_octave: int = 84

    def n_buckets(self) -> int:
        return self.buckets_per_octave * self.octaves


# ---- vqt.rs:278-348 -------------------------------------------------------------------------
@dataclass
class VqtParameters:
    sr: float = 22050.0
    n_fft: int = 2 * 16384
    range: VqtRange = field(default_factory=VqtRange)
    sparsity_quantile: float = 0.999
    quality: float = 1.6
    gamma: float = 4.8 * 1.6

    @staticmethod
    def default() -> "VqtParameters":
        p = _lib.CParams()
        _lib.load().pvq_vqt_default_params(C.byref(p))
        return VqtParameters(p.sr, p.n_fft, VqtRange(p.min_freq, p.octaves, p.buckets_per_octave),
                             p.sparsity_quantile, p.quality, p.gamma)

    def _c(self) -> _lib.CParams:
        return _lib.CParams(self.sr, self.n_fft, self.range.min_freq, self.range.octaves,
                            self.range.buckets_per_octave, self.sparsity_quantile, self.quality, self.gamma)


# ---- vqt.rs:350-366 -------------------------------------------------------------------------
class PvqError(RuntimeError):
    """Any non-OK pvq_status that is not a VqtError variant (status 9 = PVQ_ERR_NONFINITE_INPUT: a NaN / Inf sample
    reached a frame; 8 = PVQ_ERR_INTERNAL; 4 = PVQ_ERR_INVALID_ARG, also where the reference would panic)."""

    def __init__(self, status: int, msg: str):
        self.status = status
        super().__init__(f"pvq status {status}: {msg}")


class VqtError(Exception):
    pass


class AboveNyquist(VqtError):
    def __init__(self, highest_frequency: float, nyquist_frequency: float):
        self.highest_frequency, self.nyquist_frequency = highest_frequency, nyquist_frequency
        super().__init__(f"the highest VQT bin frequency ({highest_frequency} Hz) exceeds the Nyquist frequency "
                         f"({nyquist_frequency} Hz); reduce octaves or increase the sample rate")


class WindowExceedsNFft(VqtError):
    def __init__(self, window_length: float, n_fft: int):
        self.window_length, self.n_fft = window_length, n_fft
        super().__init__(f"the longest filter window ({window_length} samples) exceeds n_fft ({n_fft} samples); "
                         "increase n_fft or gamma, or decrease quality")


def last_error() -> str:
    """the calling thread's last error text (pvq_last_error: thread-local in the library)"""
    return _lib.load().pvq_last_error().decode()


def _check(st: int):
    if st != _lib.PVQ_OK:
        L = _lib.load()
        raise PvqError(st, f"{L.pvq_status_string(st).decode()}: {L.pvq_last_error().decode()}")


# ---- vqt.rs:388-415 -------------------------------------------------------------------------
@dataclass
class CsMat:
    """sprs::CsMat<Complex32> as CSR arrays."""
    shape: tuple
    indptr: np.ndarray
    indices: np.ndarray
    data: np.ndarray  # complex64

    def nnz(self) -> int:
        return int(self.indices.size)

    def rows(self) -> int:
        return self.shape[0]

    def to_dense(self) -> np.ndarray:
        d = np.zeros(self.shape, np.complex64)
        for r in range(self.shape[0]):
            s, e = self.indptr[r], self.indptr[r + 1]
            d[r, self.indices[s:e]] = self.data[s:e]
        return d


@dataclass
class WindowGroup:
    window: tuple
    filter_bank: CsMat
    negative_filter_bank: Optional[CsMat]

    def window_size(self) -> int:
        return self.window[1] - self.window[0]


@dataclass
class VqtKernel:
    window_groups: List[WindowGroup]


# ---- analysis_modules/peak_detection.rs:9-23, analysis.rs:36-98 --------------------------------
@dataclass
class PeakDetectionParameters:
    min_prominence: float
    min_height: float


@dataclass
class AnalysisParameters:
    peak_config: PeakDetectionParameters = field(default_factory=lambda: PeakDetectionParameters(10.0, 4.0))
    bassline_peak_config: PeakDetectionParameters = field(default_factory=lambda: PeakDetectionParameters(5.0, 3.5))
    highest_bassnote: int = 12 * 2 + 4
    harmonic_threshold: float = 0.3

    def _c(self) -> _lib.CAnalysisParams:
        return _lib.CAnalysisParams(self.peak_config.min_prominence, self.peak_config.min_height,
                                    self.bassline_peak_config.min_prominence, self.bassline_peak_config.min_height,
                                    self.highest_bassnote, self.harmonic_threshold)


@dataclass
class ContinuousPeak:
    center: float
    size: float


@dataclass
class FrameAnalysis:
    """Per-frame result of the stateless peak pipeline (AnalysisState::peaks, ::peaks_continuous)."""
    peaks: set
    peaks_continuous: List[ContinuousPeak]


def _ptr(t) -> int:
    """device pointer of a torch tensor (or a raw int)"""
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream_handle(stream) -> int:
    if stream is None:
        try:
            import torch
            return torch.cuda.current_stream().cuda_stream
        except Exception:
            return 0
    if isinstance(stream, int):
        return stream
    return stream.cuda_stream


class Vqt:
    """GPU-backed mirror of pitchvis_analysis::vqt::Vqt (vqt.rs:440-513, :866-916).

    `device=None` builds a host-only plan (getters work; compute raises: no CPU fallback).
    """

    def __init__(self, params: VqtParameters, device: Optional[int] = 0):
        L = _lib.load()
        self._L = L
        self._params = params
        self._h = C.c_void_p()
        err = (C.c_float * 2)()
        cp = params._c()
        st = L.pvq_vqt_create(C.byref(cp), -1 if device is None else int(device), C.byref(self._h), err)
        if st == _lib.PVQ_ERR_ABOVE_NYQUIST:
            raise AboveNyquist(err[0], err[1])
        if st == _lib.PVQ_ERR_WINDOW_EXCEEDS_NFFT:
            raise WindowExceedsNFft(err[0], int(err[1]))
        _check(st)
        self.device = device
        self.n_bins = L.pvq_vqt_n_bins(self._h)
        #: vqt.rs:449 `pub delay: Duration`, in seconds
        self.delay = L.pvq_vqt_delay_seconds(self._h)
        self.window_union = L.pvq_vqt_window_union(self._h)
        # Filter::bandwidth_3db_in_hz per bin and the kernel construction's coverage-gap warnings (vqt.rs:695-709, :956-989)
        lo = np.empty(L.pvq_vqt_n_bins(self._h), np.float32); hi = np.empty_like(lo)
        _check(L.pvq_vqt_bandwidths_3db(self._h, lo.ctypes.data_as(C.POINTER(C.c_float)), hi.ctypes.data_as(C.POINTER(C.c_float))))
        self.bandwidth_3db_in_hz = (lo, hi)
        self.warnings = []
        for i in range(L.pvq_vqt_warning_count(self._h)):
            buf = C.create_string_buffer(512)
            _check(L.pvq_vqt_warning(self._h, i, buf, 512))
            self.warnings.append(buf.value.decode())

    @classmethod
    def new(cls, params: VqtParameters, device: Optional[int] = 0) -> "Vqt":
        return cls(params, device)

    def __del__(self):
        h = getattr(self, "_h", None)
        try:
            if h is not None and h.value:
                self._L.pvq_vqt_destroy(h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    # vqt.rs:507-513
    def params(self) -> VqtParameters:
        return self._params

    def kernel(self) -> VqtKernel:
        L = self._L
        groups = []
        for g in range(L.pvq_vqt_n_groups(self._h)):
            info = (C.c_uint32 * 5)()
            _check(L.pvq_vqt_group_info(self._h, g, info))
            w0, w1, rows, nnz, nneg = [int(v) for v in info]
            mats = []
            for neg, n in ((0, nnz), (1, nneg)):
                rp = np.zeros(rows + 1, np.uint32)
                ci = np.zeros(max(n, 1), np.uint32)
                va = np.zeros(2 * max(n, 1), np.float32)
                _check(L.pvq_vqt_group_csr(self._h, g, neg, rp.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           ci.ctypes.data_as(C.POINTER(C.c_uint32)),
                                           va.ctypes.data_as(C.POINTER(C.c_float))))
                mats.append(CsMat((rows, (w1 - w0) // 2 + 1), rp, ci[:n], va[: 2 * n].view(np.complex64)))
            groups.append(WindowGroup((w0, w1), mats[0], mats[1] if nneg > 0 else None))
        return VqtKernel(groups)

    def filter_params(self):
        n = self.n_bins
        freq = np.empty(n, np.float32); wl = np.empty(n, np.float32)
        m = np.empty(n, np.uint32); mw = np.empty(n, np.uint32)
        _check(self._L.pvq_vqt_filter_params(self._h, freq.ctypes.data_as(C.POINTER(C.c_float)),
                                             wl.ctypes.data_as(C.POINTER(C.c_float)),
                                             m.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             mw.ctypes.data_as(C.POINTER(C.c_uint32))))
        return freq, wl, m, mw

    # ---- compute -----------------------------------------------------------------------------
    def calculate_vqt_instant_in_db(self, x) -> np.ndarray:
        """vqt.rs:866: x = exactly n_fft samples, the last of which is "now"."""
        x = np.ascontiguousarray(x, np.float32)
        if x.ndim != 1 or x.size != self._params.n_fft:
            # the reference panics with this message (vqt.rs:867-871)
            raise AssertionError("input must be exactly n_fft samples")
        out = np.empty(self.n_bins, np.float32)
        _check(self._L.pvq_vqt_calculate_instant_db(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.size,
                                                    out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def calculate_batch_db(self, pcm, hop: int, n_frames: Optional[int] = None, n_lead: int = 0) -> np.ndarray:
        """Host arrays in/out.  Frame f analyses the last n_fft samples after hop f (zeros before
        the stream start); pcm holds n_lead + n_frames*hop samples."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        if n_frames is None:
            n_frames = (pcm.size - n_lead) // hop
        if pcm.size < n_lead + n_frames * hop:
            raise ValueError("pcm shorter than n_lead + n_frames*hop")
        out = np.empty((n_frames, self.n_bins), np.float32)
        if n_frames:
            _check(self._L.pvq_vqt_calculate_batch_db(self._h, pcm.ctypes.data_as(C.POINTER(C.c_float)), n_lead, hop,
                                                      n_frames, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def calculate_batch_db_device(self, d_pcm, hop: int, n_frames: int, d_out_db, n_lead: int = 0,
                                  d_out_cplx=None, stream=None) -> None:
        """torch CUDA tensors (or raw device pointers); asynchronous on `stream`."""
        _check(self._L.pvq_vqt_calculate_batch_db_device(self._h, _ptr(d_pcm), n_lead, hop, n_frames, _ptr(d_out_db),
                                                         _ptr(d_out_cplx), _stream_handle(stream)))

    def analyze_batch_device(self, d_db, n_frames: int, d_peak_mask=None, d_peak_count=None, d_center=None,
                             d_size=None, max_peaks: int = 0, analysis: Optional[AnalysisParameters] = None,
                             stream=None) -> None:
        ap = (analysis or AnalysisParameters())._c()
        _check(self._L.pvq_analyze_batch_device(self._h, _ptr(d_db), n_frames, C.byref(ap), _ptr(d_peak_mask),
                                                _ptr(d_peak_count), _ptr(d_center), _ptr(d_size), max_peaks,
                                                _stream_handle(stream)))

    def vqt_analyze_batch_device(self, d_pcm, hop: int, n_frames: int, d_out_db, d_peak_mask=None, d_peak_count=None,
                                 d_center=None, d_size=None, max_peaks: int = 0, n_lead: int = 0,
                                 analysis: Optional[AnalysisParameters] = None, stream=None) -> None:
        """The whole hot path: PCM -> dB frames -> peaks, asynchronous on `stream`."""
        ap = (analysis or AnalysisParameters())._c()
        _check(self._L.pvq_vqt_analyze_batch_device(self._h, _ptr(d_pcm), n_lead, hop, n_frames, C.byref(ap),
                                                    _ptr(d_out_db), _ptr(d_peak_mask), _ptr(d_peak_count),
                                                    _ptr(d_center), _ptr(d_size), max_peaks, _stream_handle(stream)))

    def batch_streams_device(self, d_pcms, hop: int, n_frames, d_out_db, out_stride_frames: Optional[int] = None, n_leads=None,
                             d_peak_mask=None, d_peak_count=None, d_center=None, d_size=None, max_peaks: int = 0,
                             analysis: Optional[AnalysisParameters] = None, stream=None) -> None:
        """MANY streams in one call (pvq_vqt_calculate_batch_db_streams / pvq_vqt_analyze_batch_streams): d_pcms a sequence of device
        tensors (or raw pointers), n_frames a sequence; stream s's frame f goes to d_out_db[s, f] ([n_streams][out_stride_frames]
        [n_bins], the layout AnalysisBatch.preprocess_device reads).  With any peak output given the per-frame peak pipeline runs
        behind the transform, outputs laid out by the same rows.  Asynchronous on `stream`."""
        n = len(d_pcms)
        n_frames = [int(x) for x in n_frames]
        if len(n_frames) != n:
            raise ValueError("one frame count per stream")
        stride = int(out_stride_frames) if out_stride_frames is not None else (max(n_frames) if n else 0)
        ptrs = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in d_pcms])
        nf = (C.c_size_t * max(n, 1))(*n_frames)
        nl = (C.c_size_t * max(n, 1))(*[int(x) for x in n_leads]) if n_leads is not None else None
        if d_peak_mask is None and d_peak_count is None and d_center is None:
            _check(self._L.pvq_vqt_calculate_batch_db_streams(self._h, ptrs, nl, nf, n, hop, _ptr(d_out_db), stride, _stream_handle(stream)))
        else:
            ap = (analysis or AnalysisParameters())._c()
            _check(self._L.pvq_vqt_analyze_batch_streams(self._h, ptrs, nl, nf, n, hop, C.byref(ap), _ptr(d_out_db), stride, _ptr(d_peak_mask),
                                                         _ptr(d_peak_count), _ptr(d_center), _ptr(d_size), max_peaks, _stream_handle(stream)))

    @staticmethod
    def analyze_batch_multi(handles, pcm, hop: int, n_frames: int, n_lead: int = 0, analysis: Optional[AnalysisParameters] = None,
                            max_peaks: int = 64, want_peaks: bool = True):
        """One HOST stream on several handles at once (pvq_vqt_analyze_batch_multi: one host thread per handle, contiguous frame
        ranges with halos, no collective): -> (db [n_frames][n_bins], mask, count, center, size) host arrays (peak arrays None
        without want_peaks).  Bit for bit what one handle computes for the whole stream."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        if pcm.size != n_lead + n_frames * hop:
            raise ValueError("pcm must hold n_lead + n_frames * hop samples")
        v0 = handles[0]
        L = v0._L
        nb, words = v0.n_bins, (v0.n_bins + 31) // 32
        db = np.empty((n_frames, nb), np.float32)
        mask = np.zeros((n_frames, words), np.uint32) if want_peaks else None
        count = np.zeros(n_frames, np.uint32) if want_peaks else None
        center = np.zeros((n_frames, max_peaks), np.float32) if want_peaks and max_peaks else None
        size = np.zeros((n_frames, max_peaks), np.float32) if want_peaks and max_peaks else None
        hs = (C.c_void_p * len(handles))(*[h._h for h in handles])
        ap = (analysis or AnalysisParameters())._c()
        fpt, upt = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        def p_(a, t):
            return a.ctypes.data_as(t) if a is not None else None
        _check(L.pvq_vqt_analyze_batch_multi(hs, len(handles), p_(pcm, fpt), n_lead, hop, n_frames, C.byref(ap), p_(db, fpt),
                                             p_(mask, upt), p_(count, upt), p_(center, fpt), p_(size, fpt), max_peaks if center is not None else 0))
        return db, mask, count, center, size

    def analyze_batch(self, db, analysis: Optional[AnalysisParameters] = None, max_peaks: int = 64):
        """Host dB frames [n_frames][n_bins] -> (mask u32 [n_frames][words], count, center, size)."""
        db = np.ascontiguousarray(db, np.float32)
        if db.ndim == 1:
            db = db[None, :]
        if db.shape[1] != self.n_bins:
            raise AssertionError("x_vqt.len() == range.n_buckets()")  # analysis.rs:289
        n = db.shape[0]
        words = (self.n_bins + 31) // 32
        mask = np.zeros((n, words), np.uint32)
        count = np.zeros(n, np.uint32)
        center = np.zeros((n, max_peaks), np.float32)
        size = np.zeros((n, max_peaks), np.float32)
        ap = (analysis or AnalysisParameters())._c()
        if n:
            _check(self._L.pvq_analyze_batch(self._h, db.ctypes.data_as(C.POINTER(C.c_float)), n, C.byref(ap),
                                             mask.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             count.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             center.ctypes.data_as(C.POINTER(C.c_float)),
                                             size.ctypes.data_as(C.POINTER(C.c_float)), max_peaks))
        return mask, count, center, size

    def analyze_frames(self, db, analysis: Optional[AnalysisParameters] = None, max_peaks: int = 64) -> List[FrameAnalysis]:
        """AnalysisState::preprocess's peak outputs per frame, smoothing disabled (analysis.rs:332-361)."""
        mask, count, center, size = self.analyze_batch(db, analysis, max_peaks)
        out = []
        for f in range(mask.shape[0]):
            bits = np.unpackbits(mask[f].view(np.uint8), bitorder="little")[: self.n_bins]
            peaks = set(int(i) for i in np.nonzero(bits)[0])
            k = min(int(count[f]), max_peaks)
            out.append(FrameAnalysis(peaks, [ContinuousPeak(float(center[f, j]), float(size[f, j])) for j in range(k)]))
        return out

    # ---- knobs ---------------------------------------------------------------------------------
    def set_algo(self, algo: int) -> None:
        _check(self._L.pvq_vqt_set_algo(self._h, algo))

    def set_gemm_precision(self, precision: int) -> None:
        """GEMM_F32 (fp32 MFMA, default) or GEMM_BF16X3 (fp32 operands split exactly into three bf16 terms on the bf16
        matrix cores, fp32 accumulate: same parity bars, ~10 % faster end to end)"""
        _check(self._L.pvq_vqt_set_gemm_precision(self._h, precision))

    def set_workspace_limit(self, n_bytes: int) -> None:
        """upper bound of the block-DFT path's spectrum workspace (device bytes per handle; default 1 GiB): longer batches run in sub-batches"""
        _check(self._L.pvq_vqt_set_workspace_limit(self._h, int(n_bytes)))

    def set_twiddle_fp16(self, enable: bool) -> None:
        """twiddle tables rounded to fp16, fp32 accumulation (BASELINE config 4's variant); rebuilds the tables"""
        _check(self._L.pvq_vqt_set_twiddle_fp16(self._h, int(bool(enable))))

    def blockdft_columns(self) -> int:
        return int(self._L.pvq_vqt_blockdft_columns(self._h))

    def last_algo(self) -> int:
        return self._L.pvq_vqt_last_algo(self._h)

    def resolve_algo(self, hop: int, n_frames: int) -> int:
        """the path a batch of n_frames frames at this hop takes under the current setting (pvq_vqt_resolve_algo)"""
        return self._L.pvq_vqt_resolve_algo(self._h, hop, n_frames)

    def set_profiling(self, on) -> None:
        """False / True: HIP events around every kernel launch; 2: only around the transform's main kernel"""
        _check(self._L.pvq_vqt_set_profiling(self._h, 2 if on == 2 else (1 if on else 0)))

    def last_kernel_ms(self) -> dict:
        """mean GPU ms per launch of each kernel since set_profiling(True)"""
        buf = (C.c_float * 8)()
        n = self._L.pvq_vqt_last_kernel_ms(self._h, buf, 8)
        return {self._L.pvq_vqt_kernel_name(i).decode(): buf[i] for i in range(n) if buf[i] >= 0.0}

    def last_kernel_launches(self) -> dict:
        buf = (C.c_uint32 * 8)()
        n = self._L.pvq_vqt_last_kernel_launches(self._h, buf, 8)
        return {self._L.pvq_vqt_kernel_name(i).decode(): int(buf[i]) for i in range(n) if buf[i] > 0}

    def last_frames_per_launch(self) -> int:
        return int(self._L.pvq_vqt_last_frames_per_launch(self._h))

    def last_gemm_flop(self) -> float:
        """flop the matrix instructions of the last block-DFT GEMM launch issued (0 after the FFT path)"""
        return float(self._L.pvq_vqt_last_gemm_flop(self._h))

    def last_sclk_mhz(self) -> float:
        """shader clock held inside the GEMM kernel's K loop during the last profiled launch (0: not measured)"""
        return float(self._L.pvq_vqt_last_sclk_mhz(self._h))

    def input_status(self, stream=None) -> None:
        """NaN / Inf policy for the asynchronous entry points (include/pvq.h): waits for `stream`, raises
        PvqError(PVQ_ERR_NONFINITE_INPUT) if a non-finite sample reached a frame since the last check."""
        _check(self._L.pvq_vqt_input_status(self._h, _stream_handle(stream)))


# ---- analysis.rs:35-98, 119-410: stateful per-stream analysis (host side) ---------------------------
@dataclass
class FullAnalysisParameters:
    """AnalysisParameters of the reference (analysis.rs:36-65, Default :72-98); durations in seconds."""
    spectrogram_length: int = 400
    peak_config: PeakDetectionParameters = field(default_factory=lambda: PeakDetectionParameters(10.0, 4.0))
    bassline_peak_config: PeakDetectionParameters = field(default_factory=lambda: PeakDetectionParameters(5.0, 3.5))
    highest_bassnote: int = 12 * 2 + 4
    vqt_smoothing_duration_base: float = 0.070
    vqt_smoothing_calmness_min: float = 0.6
    vqt_smoothing_calmness_max: float = 2.0
    note_calmness_smoothing_duration: float = 3.5
    scene_calmness_smoothing_duration: float = 0.8
    tuning_inaccuracy_smoothing_duration: float = 4.0
    harmonic_threshold: float = 0.3

    def _c(self) -> _lib.CAnalysisFullParams:
        ns = lambda s: int(round(s * 1e9))
        return _lib.CAnalysisFullParams(
            self.spectrogram_length, self.peak_config.min_prominence, self.peak_config.min_height,
            self.bassline_peak_config.min_prominence, self.bassline_peak_config.min_height, self.highest_bassnote,
            ns(self.vqt_smoothing_duration_base), self.vqt_smoothing_calmness_min, self.vqt_smoothing_calmness_max,
            ns(self.note_calmness_smoothing_duration), ns(self.scene_calmness_smoothing_duration),
            ns(self.tuning_inaccuracy_smoothing_duration), self.harmonic_threshold)


class AnalysisBatch:
    """AnalysisState::preprocess for MANY streams on the GPU (pvq_analysis_batch_*): one wavefront per stream walks its frames in
    order, streams in parallel; same arithmetic and operation order as the host AnalysisState, state kept between calls."""

    _FIELDS = {"x_vqt_smoothed": 0, "x_vqt_peakfiltered": 1, "x_vqt_afterglow": 2, "calmness": 3, "pitch_accuracy": 4, "pitch_deviation": 5}

    def __init__(self, range: VqtRange, n_streams: int, params: Optional["FullAnalysisParameters"] = None, device: int = 0):
        self._L = _lib.load()
        self.range, self.n_streams = range, n_streams
        self.params = params or FullAnalysisParameters()
        self.n_bins = range.octaves * range.buckets_per_octave
        self._h = C.c_void_p()
        cp = self.params._c()
        _check(self._L.pvq_analysis_batch_create(device, range.min_freq, range.octaves, range.buckets_per_octave, C.byref(cp), n_streams,
                                                 C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._L.pvq_analysis_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def update_vqt_smoothing_duration(self, new_duration: Optional[float]) -> None:
        if new_duration is None:
            _check(self._L.pvq_analysis_batch_update_vqt_smoothing_duration(self._h, 0, 0))
        else:
            _check(self._L.pvq_analysis_batch_update_vqt_smoothing_duration(self._h, 1, int(round(new_duration * 1e9))))

    def preprocess_device(self, d_db, n_frames: int, frame_time: float, outputs: Optional[dict] = None, max_peaks: int = 0,
                          frame_times=None, stream=None) -> None:
        """d_db: device tensor [n_streams][n_frames][n_bins]; outputs: {field name: device tensor} (see pvq_analysis_batch_outputs);
        frame_time in seconds, or frame_times: per-frame seconds (host sequence of n_frames).  Asynchronous."""
        o = _lib.CAnalysisBatchOutputs()
        for k, t in (outputs or {}).items():
            setattr(o, k, _ptr(t))
        o.max_peaks = max_peaks
        ft = None
        if frame_times is not None:
            ft = (C.c_uint64 * n_frames)(*[int(round(x * 1e9)) for x in frame_times])
        _check(self._L.pvq_analysis_batch_preprocess_device(self._h, _ptr(d_db), n_frames, int(round(frame_time * 1e9)), ft, C.byref(o),
                                                            _stream_handle(stream)))

    def preprocess_pcm(self, vqt: "Vqt", d_pcms, n_frames: int, hop: int, frame_time: Optional[float] = None, outputs: Optional[dict] = None,
                       max_peaks: int = 0, n_leads=None, d_db=None, stream=None) -> None:
        """PCM of every stream -> VQT dB frames -> AnalysisState::preprocess, both stages on the device in one call
        (pvq_analysis_batch_preprocess_pcm); frame_time defaults to hop / sr.  d_db (optional): [n_streams][n_frames][n_bins] to keep the frames."""
        if len(d_pcms) != self.n_streams:
            raise ValueError("one PCM stream per stream of the batch")
        o = _lib.CAnalysisBatchOutputs()
        for k, t in (outputs or {}).items():
            setattr(o, k, _ptr(t))
        o.max_peaks = max_peaks
        ptrs = (C.c_void_p * self.n_streams)(*[_ptr(t) for t in d_pcms])
        nl = (C.c_size_t * self.n_streams)(*[int(x) for x in n_leads]) if n_leads is not None else None
        ft = frame_time if frame_time is not None else hop / vqt.params().sr
        _check(self._L.pvq_analysis_batch_preprocess_pcm(self._h, vqt._h, ptrs, nl, n_frames, hop, int(round(ft * 1e9)), _ptr(d_db), C.byref(o),
                                                         _stream_handle(stream)))

    def field(self, stream_index: int, name: str) -> np.ndarray:
        out = np.empty(self.n_bins, np.float32)
        _check(self._L.pvq_analysis_batch_get_field(self._h, stream_index, AnalysisBatch._FIELDS[name], out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def scalars(self, stream_index: int):
        a, b = C.c_float(), C.c_float()
        _check(self._L.pvq_analysis_batch_get_scalars(self._h, stream_index, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)


class AnalysisState:
    """Mirror of pitchvis_analysis::analysis::AnalysisState.  The frame-to-frame recurrence (EMA smoothing,
    calmness feedback, afterglow, tuning) is sequential per stream and runs on the host; feed it the dB
    frames the GPU computed.  Result fields are properties that copy out of the native state."""

    _FIELDS = {"x_vqt_smoothed": 0, "x_vqt_peakfiltered": 1, "x_vqt_afterglow": 2, "calmness": 3,
               "pitch_accuracy": 4, "pitch_deviation": 5}

    def __init__(self, range: VqtRange, params: Optional[FullAnalysisParameters] = None):
        self._L = _lib.load()
        self.range = range
        self.params = params or FullAnalysisParameters()
        self._h = C.c_void_p()
        cp = self.params._c()
        _check(self._L.pvq_analysis_state_create(range.min_freq, range.octaves, range.buckets_per_octave, C.byref(cp),
                                                 C.byref(self._h)))
        self._n = self._L.pvq_analysis_state_n_buckets(self._h)

    new = classmethod(lambda cls, range, params=None: cls(range, params))

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._L.pvq_analysis_state_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def update_vqt_smoothing_duration(self, new_duration: Optional[float]) -> None:
        """analysis.rs:251; None disables smoothing (EMA pass-through)"""
        if new_duration is None:
            _check(self._L.pvq_analysis_state_update_vqt_smoothing_duration(self._h, 0, 0))
        else:
            _check(self._L.pvq_analysis_state_update_vqt_smoothing_duration(self._h, 1, int(round(new_duration * 1e9))))

    def preprocess(self, x_vqt, frame_time: float) -> None:
        """analysis.rs:288; frame_time in seconds"""
        x = np.ascontiguousarray(x_vqt, np.float32)
        if x.ndim != 1 or x.size != self._n:
            raise AssertionError("x_vqt.len() == self.range.n_buckets()")  # analysis.rs:289 asserts
        _check(self._L.pvq_analysis_state_preprocess(self._h, x.ctypes.data_as(C.POINTER(C.c_float)), x.size,
                                                     int(round(frame_time * 1e9))))

    def bin_to_frequency(self, bin_idx: int) -> float:
        return float(self._L.pvq_analysis_state_bin_to_frequency(self._h, bin_idx))

    def _field(self, which: int) -> np.ndarray:
        out = np.empty(self._n, np.float32)
        _check(self._L.pvq_analysis_state_get_field(self._h, which, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def __getattr__(self, name):
        f = AnalysisState._FIELDS.get(name)
        if f is None:
            raise AttributeError(name)
        return self._field(f)

    @property
    def peaks(self) -> set:
        buf = np.zeros(max(self._n, 1), np.uint32)
        k = self._L.pvq_analysis_state_get_peaks(self._h, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size)
        return set(int(v) for v in buf[:k])

    @property
    def peaks_continuous(self) -> List[ContinuousPeak]:
        ce = np.zeros(max(self._n, 1), np.float32)
        sz = np.zeros(max(self._n, 1), np.float32)
        k = self._L.pvq_analysis_state_get_peaks_continuous(self._h, ce.ctypes.data_as(C.POINTER(C.c_float)),
                                                            sz.ctypes.data_as(C.POINTER(C.c_float)), ce.size)
        return [ContinuousPeak(float(ce[i]), float(sz[i])) for i in range(k)]

    @property
    def smoothed_scene_calmness(self) -> float:
        return float(self._L.pvq_analysis_state_scene_calmness(self._h))

    @property
    def smoothed_tuning_grid_inaccuracy(self) -> float:
        return float(self._L.pvq_analysis_state_tuning_grid_inaccuracy(self._h))

from .consumers import MonoAgc, PinnedArray, Stream, calculate_color, led_frame, train_chunk_samples, train_dataset, write_npy  # noqa: E402,F401
