"""SURVEY §8f rows 2 and 3 on the GPU: the pitchvis_train dataset rows (frames from the batch path) and the
streaming front end (device-resident ring), each against the oracle's literal restatement of the reference loop."""
import time

import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from oracle import consumers as OC

pytestmark = pytest.mark.gpu


def _train_params():
    # pitchvis_train/src/train.rs:30-42
    q = 10.0
    pp = P.VqtParameters(sr=22050.0, n_fft=32768, range=P.VqtRange(55.0, 7, 36), sparsity_quantile=0.999, quality=q, gamma=5.3 * q)
    op = O.OracleParams(sr=22050.0, n_fft=32768, min_freq=55.0, octaves=7, buckets_per_octave=36, sparsity_quantile=0.999,
                        quality=q, gamma=5.3 * q)
    return pp, op


def _render(n, sr, seed):
    """a seeded piano-roll stand-in for the synthesizer: decaying partials on MIDI keys, stereo"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    left = np.zeros(n)
    right = np.zeros(n)
    notes = []
    for _ in range(10):
        key = int(rng.integers(40, 90))
        t0 = float(rng.uniform(0, t[-1] * 0.8))
        f0 = 440.0 * 2 ** ((key - 69) / 12)
        env = np.where(t >= t0, np.exp(-(t - t0) * 3.0), 0.0)
        tone = sum(np.sin(2 * np.pi * f0 * h * t) / h ** 2 for h in range(1, 5)) * env * 0.2
        pan = rng.uniform(0.2, 0.8)
        left += tone * pan
        right += tone * (1 - pan)
        notes.append((key, t0, pan))
    return left.astype(np.float32), right.astype(np.float32), notes


def test_train_dataset_vs_reference_loop():
    pp, op = _train_params()
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    chunk = P.train_chunk_samples(v)
    assert chunk == OC.train_chunk_samples(v.delay, 22050)
    n_chunks, step = 42, 3
    left, right, notes = _render(n_chunks * chunk, 22050.0, 3)
    left[10 * chunk:12 * chunk] = 0.0
    right[10 * chunk:12 * chunk] = 0.0
    n_frames = n_chunks // step
    voices = []
    for f in range(n_frames):
        tt = (f + 1) * step * chunk / 22050.0
        voices.append([(k, float(np.exp(-(tt - t0) * 3.0) * pan * 4), float(np.exp(-(tt - t0) * 3.0) * (1 - pan) * 4))
                       for k, t0, pan in notes if tt >= t0])
    rows = P.train_dataset(v, left, right, voices, step=step).reshape(n_frames, -1)
    want, _, _ = OC.train_loop(ov, left, right, voices, chunk, step)
    want = want.reshape(n_frames, -1)
    nb = v.n_bins
    assert rows.shape == (n_frames, nb + 128)
    assert np.array_equal(rows[:, nb:], want[:, nb:])            # targets: exact
    assert rows[:, nb:].sum() > 0
    # dB values: the batch GPU frames against one reference-style call per analysed chunk
    err = np.abs(rows[:, :nb] - want[:, :nb])
    loud = want[:, :nb] > 1.0
    assert err[loud].max() <= 2e-2 and np.median(err[loud]) <= 2e-4, (err[loud].max(), np.median(err[loud]))
    assert err.max() <= 0.5                                       # bins at the 60 dB floor carry the frame's cancellation noise


def test_stream_ring_and_frames():
    pp = P.VqtParameters(sr=22050.0, n_fft=32768, range=P.VqtRange(55.0, 5, 36), sparsity_quantile=0.999, quality=1.8,
                         gamma=4.8 * 1.8)                        # pitchvis_serial/src/main.rs:19-42
    op = O.OracleParams(sr=22050.0, n_fft=32768, min_freq=55.0, octaves=5, buckets_per_octave=36, sparsity_quantile=0.999,
                        quality=1.8, gamma=4.8 * 1.8)
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    bufsize = 2 * 22050                                          # main.rs:22
    s = P.Stream(v, bufsize, with_agc=True)
    agc = OC.MonoAgc(0.07, 0.0001)                               # audio_desktop.rs:93
    ring = np.zeros(bufsize, np.float32)
    assert np.array_equal(s.read(), ring) and s.gain == 0.0
    rng = np.random.default_rng(5)
    t = 0
    sizes = [441, 1024, 37, 2048, 512] * 60                      # > 3 x bufsize in total: the device ring compacts
    lat = []
    for i, n in enumerate(sizes):
        tt = (t + np.arange(n)) / 22050.0
        data = (0.3 * np.sin(2 * np.pi * 220.0 * tt) + 0.1 * np.sin(2 * np.pi * 1234.5 * tt) + 0.01 * rng.standard_normal(n)).astype(np.float32)
        if i % 17 == 5:
            data[:] = 0.0                                        # silence: gain frozen (audio_desktop.rs:101-102)
        t += n
        bad = data.copy()
        bad[n // 2] = np.nan
        s.push(bad)                                              # dropped whole (audio_desktop.rs:97-100)
        t0 = time.perf_counter()
        s.push(data)
        if i % 10 == 9:
            db = s.frame_db()
            lat.append(time.perf_counter() - t0)
        # the callback, literally (audio_desktop.rs:101-118)
        sq = np.float32(0.0)
        for x in data:
            sq = np.float32(sq + np.float32(x * x))
        agc.freeze_gain(sq < np.float32(1e-6))
        ring = np.concatenate([ring[n:], data])
        new = ring[-n:].copy()
        agc.process(new)
        ring[-n:] = new
        assert s.gain == float(agc.gain)
        assert abs(s.chunk_size_ms - n / 22050.0 * 1000.0) < 1e-3
        if i % 10 == 9:
            assert np.array_equal(s.read(), ring)                # bit-exact ring contents
            want = ov.calculate_vqt_instant_in_db(ring[-32768:]) # pitchvis_serial/src/main.rs:205-211
            err = np.abs(db - want)
            loud = want > 1.0
            assert err[loud].max() <= 2e-2 and err.max() <= 0.5
    print(f"stream push + frame latency: median {np.median(lat) * 1e6:.0f} us, min {np.min(lat) * 1e6:.0f} us")
    with pytest.raises(P.PvqError):
        s.push(np.zeros(bufsize + 1, np.float32))
    with pytest.raises(P.PvqError):
        P.Stream(v, 1000)


def test_pinned_host_buffers():
    """pvq_host_alloc: page-locked buffers for the host-buffer entry points give the same bits as pageable ones"""
    pp, _ = _train_params()
    v = P.Vqt.new(pp, 0)
    hop, nf = 256, 300
    pcm = (np.random.default_rng(3).standard_normal(hop * nf) * 0.1).astype(np.float32)
    want = v.calculate_batch_db(pcm, hop, nf)
    pin = P.PinnedArray((hop * nf,))
    pin.array[:] = pcm
    got = v.calculate_batch_db(pin.array, hop, nf)
    assert np.array_equal(got, want)
    del pin
    # a batch large enough for the three-stream upload / run / download pipeline: same bits as the device entry point
    import torch
    nf = 40000
    pcm = (np.random.default_rng(4).standard_normal(1000 + hop * nf) * 0.1).astype(np.float32)
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(torch.from_numpy(pcm).cuda(), hop, nf, d_db, n_lead=1000)
    torch.cuda.synchronize()
    want = d_db.cpu().numpy()
    assert np.array_equal(v.calculate_batch_db(pcm, hop, nf, n_lead=1000), want)
    pin = P.PinnedArray((pcm.size,))
    pin.array[:] = pcm
    assert np.array_equal(v.calculate_batch_db(pin.array, hop, nf, n_lead=1000), want)


def test_serial_led_frames_from_batched_gpu_peaks():
    """SURVEY 8f row 4 on the GPU: pitchvis_serial's loop (main.rs:198-220: ring buffer -> VQT -> AnalysisState::preprocess ->
    update_serial) as a batch.  The serial geometry (main.rs:17-39: 22 050 Hz, 5 x 36 bins, Q 1.8), one frame per 1/30 s.
    (a) stateless: pvq_vqt_analyze_batch_device's peaks_continuous -> pvq_led_frame, against oracle/consumers.py's
        update_serial restatement fed with the oracle's peak pipeline on the same GPU dB frame: the same bytes, except that a
        (size / max) * 254 product sitting on an integer boundary may quantise one level apart (centre / size carry the
        1e-4-bin / 2e-3-dB tolerance of tests/test_peaks_gpu.py);
    (b) stateful, as the reference runs it: GPU dB frames -> the product's host AnalysisState -> pvq_led_frame, against
        oracle/analysis_state.py -> oracle update_serial."""
    torch = pytest.importorskip("torch")
    from oracle.analysis_state import OracleAnalysisState
    from pitchvis_amd import consumers as PC
    from helpers import get_geom, white_noise
    from synth import piano_roll
    pp, op = get_geom("serial_22k_180")
    v = P.Vqt.new(pp, 0)
    n, bpo = v.n_bins, op.buckets_per_octave
    pcm, _ = piano_roll(op.sr, 8.0, 21)
    pcm = (pcm * 2.0 + white_noise(pcm.size, 21, amp=0.004)).astype(np.float32)
    hop = int(op.sr) // 30                                   # FPS = 30 (main.rs:41); 735 samples: the FFT path
    nf = pcm.size // hop
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((nf, n), device="cuda")
    d_mask = torch.zeros((nf, (n + 31) // 32), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
    d_c = torch.zeros((nf, 64), device="cuda")
    d_s = torch.zeros((nf, 64), device="cuda")
    v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64)
    torch.cuda.synchronize()
    db, cnt, ctr, sz = d_db.cpu().numpy(), d_cnt.cpu().numpy(), d_c.cpu().numpy(), d_s.cpu().numpy()
    assert cnt.max() >= 3 and (cnt > 0).sum() > nf // 2       # the stimulus really lights LEDs
    exact = off_by_one = 0
    for f in range(nf):
        k = int(cnt[f])
        got = P.led_frame(n, bpo, list(zip(ctr[f, :k].tolist(), sz[f, :k].tolist())))
        _, wce, wsz = O.analyze_frame(db[f], op.min_freq, op.octaves, bpo)
        want = OC.led_frame(n, bpo, list(zip(wce.tolist(), wsz.tolist())), PC.SERIAL_COLORS, PC.SERIAL_GRAY_LEVEL,
                            PC.SERIAL_EASING_POW)
        assert len(got) == len(want) == 3 + 3 * n and got[:3] == want[:3] and max(got[3:]) <= 0xFE
        d = np.abs(np.frombuffer(got, np.uint8).astype(int) - np.frombuffer(want, np.uint8).astype(int))
        assert d.max() <= 1, f
        exact += int(d.max() == 0)
        off_by_one += int(d.max() == 1)
    assert exact >= int(0.97 * nf), (exact, off_by_one, nf)
    # (b) the reference's stateful loop
    st = P.AnalysisState.new(pp.range)
    ost = OracleAnalysisState(op.min_freq, op.octaves, bpo)
    same = 0
    for f in range(nf):
        st.preprocess(db[f], 1.0 / 30.0)
        ost.preprocess(db[f], int(round(1e9 / 30.0)))
        got = P.led_frame(n, bpo, [(p.center, p.size) for p in st.peaks_continuous])
        want = OC.led_frame(n, bpo, list(zip(ost.centers.tolist(), ost.sizes.tolist())), PC.SERIAL_COLORS,
                            PC.SERIAL_GRAY_LEVEL, PC.SERIAL_EASING_POW)
        assert sorted(st.peaks) == list(ost.peaks), f
        d = np.abs(np.frombuffer(got, np.uint8).astype(int) - np.frombuffer(want, np.uint8).astype(int))
        # the two EMA evaluations differ by an ulp and the log-frequency parabola amplifies that to ~1e-2 bins
        # (tests/test_analysis_state.py): the brightness split between the two buckets of a peak moves by a few levels
        assert d.max() <= 6, (f, int(d.max()))
        same += int(d.max() <= 1)
    assert same >= int(0.9 * nf), (same, nf)
    print(f"LED frames: stateless {exact}/{nf} identical, {off_by_one} one level apart; stateful {same}/{nf} within one level")
