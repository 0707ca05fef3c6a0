"""Callers either side of the VQT path, mirroring the reference's own items:

* ``MonoAgc``                        — dagc_fork/src/lib.rs:19-87
* ``train_dataset`` / ``write_npy``  — pitchvis_train/src/train.rs:252-351, 443-460, 192-208 (frames on the GPU)
* ``Stream``                         — the pitchvis_audio RingBuffer contract with a device-resident ring
* ``calculate_color`` / ``led_frame``— pitchvis_colors/src/lib.rs:86-117, pitchvis_serial/src/main.rs:122-175
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib

_fp = C.POINTER(C.c_float)


def _f(a: np.ndarray):
    return a.ctypes.data_as(_fp)


def _check(st: int):
    if st != _lib.PVQ_OK:
        L = _lib.load()
        from . import PvqError
        raise PvqError(st, (L.pvq_last_error() or b"").decode())


class MonoAgc:
    """dagc::MonoAgc (dagc_fork/src/lib.rs:19-87)"""

    def __init__(self, desired_output_rms: float, distortion_factor: float):
        self._L = _lib.load()
        self._h = C.c_void_p()
        st = self._L.pvq_mono_agc_create(desired_output_rms, distortion_factor, C.byref(self._h))
        if st != _lib.PVQ_OK:
            raise ValueError((self._L.pvq_last_error() or b"").decode())

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.pvq_mono_agc_destroy(h)
            self._h = None

    def freeze_gain(self, freeze: bool) -> None:
        self._L.pvq_mono_agc_freeze_gain(self._h, int(bool(freeze)))

    def is_gain_frozen(self) -> bool:
        return bool(self._L.pvq_mono_agc_is_gain_frozen(self._h))

    def gain(self) -> float:
        return float(self._L.pvq_mono_agc_gain(self._h))

    def process(self, samples: np.ndarray) -> None:
        """in place, like the reference (lib.rs:76)"""
        assert samples.dtype == np.float32 and samples.flags.c_contiguous
        self._L.pvq_mono_agc_process(self._h, _f(samples), samples.size)


STEP_SIZE_IN_CHUNKS = 3   # train.rs:44


def train_chunk_samples(vqt) -> int:
    """train.rs:128-129"""
    return int(_lib.load().pvq_train_chunk_samples(vqt._h))


def train_dataset(vqt, left: np.ndarray, right: Optional[np.ndarray], voices: Sequence[Sequence[Tuple[int, float, float]]],
                  step: int = STEP_SIZE_IN_CHUNKS, agc: Optional[MonoAgc] = None, chunk: Optional[int] = None) -> np.ndarray:
    """The body of pitchvis_train::synthesize_midi_to_wav + generate_data for one rendered stream
    (train.rs:252-351, 443-460).  ``left``/``right``: the synthesizer's output, a whole number of chunks;
    ``voices[f]``: (key, mix_gain_left, mix_gain_right) of the voices sounding at analysed chunk f.
    Returns the flat float32 rows ``[n_frames * (n_bins + 128)]`` that train() concatenates into data.npy."""
    L = _lib.load()
    chunk = train_chunk_samples(vqt) if chunk is None else int(chunk)
    left = np.ascontiguousarray(left, np.float32)
    n_chunks = left.size // chunk
    assert n_chunks * chunk == left.size, "the stream must be a whole number of chunks"
    if right is not None:
        right = np.ascontiguousarray(right, np.float32)
        assert right.size == left.size
    agc = agc or MonoAgc(0.07, 0.001)   # train.rs:265
    mono = np.empty(n_chunks * chunk, np.float32)
    gains = np.empty(n_chunks, np.float32)
    _check(L.pvq_train_condition_stream(agc._h, _f(left), _f(right) if right is not None else None, n_chunks, chunk,
                                        _f(mono), _f(gains)))
    n_frames = n_chunks // step
    db = np.empty((n_frames, vqt.n_bins), np.float32)
    _check(L.pvq_train_frames_db(vqt._h, _f(mono), n_chunks, chunk, step, _f(db)))
    assert len(voices) == n_frames
    ptr = np.zeros(n_frames + 1, np.uint32)
    keys, gl, gr = [], [], []
    for f, vs in enumerate(voices):
        for k, a, b in vs:
            keys.append(k); gl.append(a); gr.append(b)
        ptr[f + 1] = len(keys)
    keys = np.asarray(keys if keys else [0], np.int32)
    gl = np.asarray(gl if gl else [0], np.float32)
    gr = np.asarray(gr if gr else [0], np.float32)
    agc_gain = np.ascontiguousarray(gains[step - 1::step][:n_frames])   # agc.gain() at the analysed chunks (train.rs:326)
    rows = np.empty((n_frames, vqt.n_bins + 128), np.float32)
    _check(L.pvq_train_rows(_f(db), n_frames, vqt.n_bins, ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                            keys.ctypes.data_as(C.POINTER(C.c_int32)), _f(gl), _f(gr), _f(agc_gain), _f(rows)))
    return rows.reshape(-1)


def write_npy(path: str, data: np.ndarray) -> None:
    """train.rs:192-208: flat '<f4' .npy"""
    data = np.ascontiguousarray(data, np.float32).reshape(-1)
    _check(_lib.load().pvq_npy_write_f32(str(path).encode(), _f(data), data.size))


class Stream:
    """pitchvis_audio::RingBuffer (lib.rs:17-22) fed like audio_desktop.rs:88-131, ring on the device"""

    def __init__(self, vqt, buf_size: int, with_agc: bool = True):
        self._L = _lib.load()
        self._vqt = vqt
        self._h = C.c_void_p()
        self.buf_size = int(buf_size)
        _check(self._L.pvq_stream_create(vqt._h, buf_size, int(with_agc), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.pvq_stream_destroy(h)
            self._h = None

    def push(self, data: np.ndarray) -> None:
        data = np.ascontiguousarray(data, np.float32)
        _check(self._L.pvq_stream_push(self._h, _f(data), data.size))

    @property
    def gain(self) -> float:
        return float(self._L.pvq_stream_gain(self._h))

    @property
    def chunk_size_ms(self) -> float:
        return float(self._L.pvq_stream_chunk_size_ms(self._h))

    def frame_db(self) -> np.ndarray:
        out = np.empty(self._vqt.n_bins, np.float32)
        _check(self._L.pvq_stream_frame_db(self._h, _f(out)))
        return out

    def read(self, n_last: Optional[int] = None) -> np.ndarray:
        n = self.buf_size if n_last is None else int(n_last)
        out = np.empty(n, np.float32)
        _check(self._L.pvq_stream_read(self._h, _f(out), n))
        return out


class PinnedArray:
    """float32 NumPy view of page-locked host memory (pvq_host_alloc): hand `.array` to the host-buffer entry points"""

    def __init__(self, shape):
        self._L = _lib.load()
        n = int(np.prod(shape))
        self._p = self._L.pvq_host_alloc(4 * max(n, 1))
        if not self._p:
            raise MemoryError((self._L.pvq_last_error() or b"").decode())
        buf = (C.c_float * max(n, 1)).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=np.float32, count=n).reshape(shape)

    def __del__(self):
        p = getattr(self, "_p", None)
        if p:
            self.array = None
            self._L.pvq_host_free(p)
            self._p = None


# pitchvis_colors/src/lib.rs:19-36
COLORS = np.array([
    [0.85, 0.36, 0.36], [0.01, 0.52, 0.71], [0.97, 0.76, 0.05], [0.45, 0.34, 0.63], [0.47, 0.77, 0.22], [0.78, 0.32, 0.52],
    [0.00, 0.64, 0.56], [0.95, 0.54, 0.23], [0.30, 0.37, 0.64], [1.00, 0.96, 0.03], [0.57, 0.30, 0.55], [0.12, 0.71, 0.34],
], np.float32)
GRAY_LEVEL, EASING_POW = 60.0, 1.3                       # lib.rs:56-57
# pitchvis_serial/src/main.rs:44-59
SERIAL_COLORS = np.array([
    [0.95, 0.10, 0.10], [0.01, 0.52, 0.71], [0.97, 0.79, 0.00], [0.45, 0.34, 0.63], [0.47, 0.99, 0.02], [0.88, 0.02, 0.52],
    [0.00, 0.80, 0.55], [0.99, 0.54, 0.03], [0.25, 0.30, 0.64], [0.95, 0.99, 0.00], [0.52, 0.00, 0.60], [0.05, 0.80, 0.15],
], np.float32)
SERIAL_GRAY_LEVEL, SERIAL_EASING_POW = 5.0, 2.3


def calculate_color(buckets_per_octave: int, bucket: float, colors: np.ndarray = COLORS, gray_level: float = GRAY_LEVEL,
                    easing_pow: float = EASING_POW) -> Tuple[float, float, float]:
    colors = np.ascontiguousarray(colors, np.float32)
    out = np.empty(3, np.float32)
    _lib.load().pvq_calculate_color(buckets_per_octave, bucket, _f(colors), gray_level, easing_pow, _f(out))
    return float(out[0]), float(out[1]), float(out[2])


def led_frame(n_buckets: int, buckets_per_octave: int, peaks_continuous: Sequence[Tuple[float, float]],
              colors: np.ndarray = SERIAL_COLORS, gray_level: float = SERIAL_GRAY_LEVEL,
              easing_pow: float = SERIAL_EASING_POW) -> bytes:
    """pitchvis_serial::update_serial (main.rs:122-175): the bytes written to the serial port"""
    colors = np.ascontiguousarray(colors, np.float32)
    ctr = np.asarray([p[0] for p in peaks_continuous] or [0.0], np.float32)
    sz = np.asarray([p[1] for p in peaks_continuous] or [0.0], np.float32)
    out = np.zeros(3 + 3 * n_buckets, np.uint8)
    n = _lib.load().pvq_led_frame(n_buckets, buckets_per_octave, _f(ctr), _f(sz), len(peaks_continuous), _f(colors), gray_level,
                                  easing_pow, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out[:n].tobytes()
