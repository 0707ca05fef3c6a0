"""SURVEY §8f rows 2 and 4 on the CPU: MonoAgc, the pitchvis_train conditioning / row / .npy writer, colour
mapping and the serial LED frame — product (C ABI, host code) against the NumPy-f32 oracle restatements and
the reference's own tests for these items."""
import math
import os

import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from oracle import consumers as OC
from pitchvis_amd import consumers as PC


def test_mono_agc_reference_it_works():
    # dagc_fork/src/lib.rs:93-109
    agc = P.MonoAgc(0.001, 0.0001)
    assert agc.gain() == 1.0
    assert not agc.is_gain_frozen()
    agc.freeze_gain(True)
    assert agc.is_gain_frozen()
    samples = np.array([0.5, 1.0, -0.2], np.float32)
    agc.process(samples)
    assert agc.gain() == 1.0
    agc.freeze_gain(False)
    agc.process(samples)
    assert agc.gain() != 1.0


@pytest.mark.parametrize("rms,dist", [(0.0, 0.001), (-1.0, 0.001), (float("inf"), 0.001), (float("nan"), 0.1), (0.07, -0.1),
                                      (0.07, 1.5), (0.07, float("nan"))])
def test_mono_agc_rejects_like_the_reference(rms, dist):
    # lib.rs:36-49
    with pytest.raises(ValueError) as e:
        P.MonoAgc(rms, dist)
    with pytest.raises(ValueError) as eo:
        OC.MonoAgc(rms, dist)
    assert ("desired_output_rms" in str(e.value)) == ("desired_output_rms" in str(eo.value))


@pytest.mark.filterwarnings("ignore::RuntimeWarning")
def test_mono_agc_bit_exact_vs_oracle():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(6000) * 0.2).astype(np.float32)
    x[1000:1500] = 0.0
    x[3000] = 40.0   # drives g to the distortion-factor floor (lib.rs:81)
    for rms, d in ((0.07, 0.001), (0.07, 0.0001), (0.5, 1.0), (0.01, 0.0)):
        a, o = P.MonoAgc(rms, d), OC.MonoAgc(rms, d)
        xa, xo = x.copy(), x.copy()
        for lo in range(0, x.size, 750):
            frozen = lo == 1500
            a.freeze_gain(frozen); o.freeze_gain(frozen)
            seg_a, seg_o = xa[lo:lo + 750], xo[lo:lo + 750]
            a.process(seg_a); o.process(seg_o)
            assert a.gain() == float(o.gain) or (math.isnan(a.gain()) and math.isnan(float(o.gain)))
        assert np.array_equal(xa, xo, equal_nan=True)


def _train_vqt(device=None):
    # pitchvis_train/src/train.rs:30-42
    q = 10.0
    return P.VqtParameters(sr=22050.0, n_fft=32768, range=P.VqtRange(55.0, 7, 36), sparsity_quantile=0.999, quality=q,
                           gamma=5.3 * q)


def test_train_chunk_samples():
    v = P.Vqt.new(_train_vqt(), None)
    chunk = P.train_chunk_samples(v)
    assert chunk == OC.train_chunk_samples(v.delay, 22050)
    assert chunk % 64 == 0 and 0 < chunk <= int(v.delay * 22050)


def test_condition_stream_and_rows_vs_oracle(tmp_path):
    rng = np.random.default_rng(11)
    chunk, n_chunks, step = 192, 30, 3
    left = (rng.standard_normal(chunk * n_chunks) * 0.1).astype(np.float32)
    right = (rng.standard_normal(chunk * n_chunks) * 0.1).astype(np.float32)
    left[5 * chunk:7 * chunk] = 0.0
    right[5 * chunk:7 * chunk] = 0.0          # two silent chunks: the gain freezes (train.rs:292-293)
    L = P._lib.load()
    import ctypes as C
    fp = C.POINTER(C.c_float)
    agc = P.MonoAgc(0.07, 0.001)
    mono = np.empty_like(left)
    gains = np.empty(n_chunks, np.float32)
    assert L.pvq_train_condition_stream(agc._h, left.ctypes.data_as(fp), right.ctypes.data_as(fp), n_chunks, chunk,
                                        mono.ctypes.data_as(fp), gains.ctypes.data_as(fp)) == 0
    # oracle: the literal loop with a stub transform (the frames are checked on the GPU)
    class Stub:
        class params:
            n_fft, sr = 512, 22050.0
        def calculate_vqt_instant_in_db(self, x):
            return np.asarray(x[-8:], np.float32)
    n_frames = n_chunks // step
    voices = [[(60 + (f % 5), 0.9, 0.7), (60 + (f % 5), 0.2, 0.1), (72, 0.4 * (f % 3), 0.4)] for f in range(n_frames)]
    rows_o, gains_o, ring_o = OC.train_loop(Stub(), left, right, voices, chunk, step, bufsize=chunk * n_chunks)
    assert np.array_equal(gains, gains_o)
    assert np.array_equal(mono, ring_o)                       # the ring holds the whole conditioned stream here
    assert gains[5] == gains[6] == gains[4]                   # frozen over the silent chunks
    # rows
    db = np.stack([mono[(f + 1) * step * chunk - 8:(f + 1) * step * chunk] for f in range(n_frames)]).astype(np.float32)
    ptr = np.zeros(n_frames + 1, np.uint32)
    keys, gl, gr = [], [], []
    for f, vs in enumerate(voices):
        for k, a, b in vs:
            keys.append(k); gl.append(a); gr.append(b)
        ptr[f + 1] = len(keys)
    keys, gl, gr = np.asarray(keys, np.int32), np.asarray(gl, np.float32), np.asarray(gr, np.float32)
    ag = np.ascontiguousarray(gains[step - 1::step])
    rows = np.empty((n_frames, 8 + 128), np.float32)
    assert L.pvq_train_rows(db.ctypes.data_as(fp), n_frames, 8, ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                            keys.ctypes.data_as(C.POINTER(C.c_int32)), gl.ctypes.data_as(fp), gr.ctypes.data_as(fp),
                            ag.ctypes.data_as(fp), rows.ctypes.data_as(fp)) == 0
    assert np.array_equal(rows.reshape(-1), rows_o)
    assert rows[0, 8:].sum() == 0                              # frame 0 is labelled with the empty previous set (train.rs:314,347)
    assert rows[1:, 8:].sum() > 0
    # a key outside 0..127 indexes out of the reference's [f32; 128]: refused
    keys_bad = keys.copy(); keys_bad[0] = 128
    assert L.pvq_train_rows(db.ctypes.data_as(fp), n_frames, 8, ptr.ctypes.data_as(C.POINTER(C.c_uint32)),
                            keys_bad.ctypes.data_as(C.POINTER(C.c_int32)), gl.ctypes.data_as(fp), gr.ctypes.data_as(fp),
                            ag.ctypes.data_as(fp), rows.ctypes.data_as(fp)) == P._lib.PVQ_ERR_INVALID_ARG
    # .npy (train.rs:192-208): flat '<f4'
    path = os.path.join(tmp_path, "data.npy")
    P.write_npy(path, rows)
    back = np.load(path)
    assert back.dtype == np.dtype("<f4") and back.shape == (rows.size,) and np.array_equal(back, rows.reshape(-1))
    with open(path, "rb") as fh:
        head = fh.read(10)
    assert head[:6] == b"\x93NUMPY" and head[6:8] == b"\x01\x00" and (10 + int.from_bytes(head[8:10], "little")) % 64 == 0
    P.write_npy(path, np.zeros(0, np.float32))
    assert np.load(path).shape == (0,)


@pytest.mark.parametrize("palette,gray,ease", [(PC.COLORS, PC.GRAY_LEVEL, PC.EASING_POW),
                                                (PC.SERIAL_COLORS, PC.SERIAL_GRAY_LEVEL, PC.SERIAL_EASING_POW)])
def test_calculate_color(palette, gray, ease):
    for bpo in (12, 36, 84):
        per = bpo // 12
        for tone in range(12):
            # on a tone the colour is the palette entry quantised to u8 (pitchvis_colors/src/lib.rs:77-79, 94-95)
            got = P.calculate_color(bpo, float(tone * per), palette, gray, ease)
            want = tuple(int(np.float32(c) * np.float32(255.0)) / 255.0 for c in palette[tone])
            assert max(abs(a - b) for a, b in zip(got, want)) <= 1.0 / 255.0 + 1e-6
            # half way between two tones the chroma is gone: a gray of lightness gray_level (lib.rs:104-106)
            if per % 2 == 0:
                r, g, b = P.calculate_color(bpo, tone * per + per / 2, palette, gray, ease)
                assert max(r, g, b) - min(r, g, b) <= 2.0 / 255.0
        for b in np.linspace(0.0, bpo, 61, endpoint=False):
            got = P.calculate_color(bpo, float(np.float32(b)), palette, gray, ease)
            want = OC.calculate_color(bpo, float(np.float32(b)), palette, gray, ease)
            assert max(abs(a - w) for a, w in zip(got, want)) <= 1.0 / 255.0 + 1e-6   # libm pow/cbrt may differ by an ulp


def test_led_frame():
    n, bpo = 180, 36          # pitchvis_serial: 5 octaves x 36 (main.rs:24-27)
    peaks = [(12.25, 7.5), (48.0, 3.0), (100.9, 9.0), (179.4, 2.0)]
    out = P.led_frame(n, bpo, peaks)
    assert out == OC.led_frame(n, bpo, peaks, PC.SERIAL_COLORS, PC.SERIAL_GRAY_LEVEL, PC.SERIAL_EASING_POW)
    assert len(out) == 3 + 3 * n and out[0] == 0xFF and out[1] * 256 + out[2] == n      # main.rs:146-150
    assert max(out[3:]) <= 0xFE                                                         # 0xFF only marks the start
    lit = {i for i in range(n) if any(out[3 + 3 * i:6 + 3 * i])}
    assert lit <= {12, 13, 48, 100, 101, 179} and {12, 48, 101} <= lit                  # main.rs:131-140
    # the strongest bucket shows its colour at full strength: (c * 254) as u8
    x = np.zeros(n, np.float32)
    for c, s in peaks:
        fr = np.float32(c) - np.floor(np.float32(c))
        x[int(c)] = np.float32(s) * (1 - np.float32(fr) ** np.float32(1.9))
        if int(c) < n - 1:
            x[int(c) + 1] = np.float32(s) * np.float32(fr) ** np.float32(1.9)
    k = int(np.argmax(x))
    shift = bpo - 3 * (bpo // 12)
    rgb = P.calculate_color(bpo, float((k + shift) % bpo), PC.SERIAL_COLORS, PC.SERIAL_GRAY_LEVEL, PC.SERIAL_EASING_POW)
    assert list(out[3 + 3 * k:6 + 3 * k]) == [int(np.float32(v) * np.float32(254.0)) for v in rgb]
    # no peaks: max_size 0 -> 0/0 -> NaN -> `as u8` 0 (main.rs:142-143,162-167)
    dark = P.led_frame(n, bpo, [])
    assert dark[:3] == out[:3] and not any(dark[3:])
