"""AnalysisState::preprocess for many streams on the GPU (pvq_analysis_batch_*, SURVEY.md 8f row 1's "one wave per stream" half)
against the product's host AnalysisState — itself checked frame by frame against oracle/analysis_state.py and the reference's own
tests for this layer (tests/test_analysis_state.py) — in the reference's DEFAULT smoothing mode (70 ms calmness-adaptive EMA,
analysis.rs:72-98), 256 streams x 1 000 frames, every pub field of every frame.

What is compared how: the peak index sets (bit masks) and counts per frame must be EQUAL in every frame of every stream that is
compared in full; the float fields — per-bin EMAs, afterglow, calmness, peak-filtered frame, pitch accuracy / deviation,
peaks_continuous, scene calmness, tuning inaccuracy — are compared BIT FOR BIT and the share of exactly equal values is reported;
the few that differ (a libm call rounded the other way: the GPU evaluates exp / ln / log2 / powf in double and rounds once, glibc's
f32 routines are correctly rounded in all but ~1e-8 of their calls, log10f less often) must stay within the tolerances
tests/test_analysis_state.py uses between the host state and the NumPy oracle."""
import numpy as np
import pytest

import pitchvis_amd as P

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FIELDS = ("x_vqt_smoothed", "x_vqt_peakfiltered", "x_vqt_afterglow", "calmness", "pitch_accuracy", "pitch_deviation")


def _frames(n_streams, n_frames, n_bins, seed):
    """dB-like frames: a noise floor, a few notes per stream that start, hold, glide and stop, a silent stretch"""
    rng = np.random.default_rng(seed)
    x = (rng.random((n_streams, n_frames, n_bins), dtype=np.float32) * 6.0).astype(np.float32)
    t = np.arange(n_frames)
    for s in range(n_streams):
        for _ in range(int(rng.integers(2, 7))):
            b0 = int(rng.integers(3, n_bins - 3))
            t0 = int(rng.integers(0, n_frames - 50))
            t1 = min(n_frames, t0 + int(rng.integers(30, 400)))
            glide = rng.integers(-1, 2) * (t[t0:t1] - t0) // 97
            lvl = float(rng.uniform(18.0, 50.0))
            for dt, b in zip(range(t0, t1), np.clip(b0 + glide, 2, n_bins - 3)):
                x[s, dt, b] = lvl + 0.3 * np.sin(dt / 7.0)
                x[s, dt, b - 1] = max(x[s, dt, b - 1], lvl - 9.0)
                x[s, dt, b + 1] = max(x[s, dt, b + 1], lvl - 11.0)
        q = int(rng.integers(100, n_frames - 100))
        x[s, q:q + 25] = 0.0
    return x


def _host_reference(rng_, x, dt, mode):
    """the product's host AnalysisState over one stream: every pub field per frame"""
    st = P.AnalysisState.new(rng_)
    if mode == "none":
        st.update_vqt_smoothing_duration(None)
    elif mode == "retuned":
        st.update_vqt_smoothing_duration(0.120)
    nf, nb = x.shape
    out = {k: np.empty((nf, nb), np.float32) for k in FIELDS}
    out["mask"] = np.zeros((nf, nb), bool)
    out["scene"] = np.empty(nf, np.float32); out["tuning"] = np.empty(nf, np.float32)
    out["pc"] = []
    for f in range(nf):
        st.preprocess(x[f], dt)
        for k in FIELDS:
            out[k][f] = getattr(st, k)
        for p in st.peaks:
            out["mask"][f, p] = True
        out["pc"].append([(c.center, c.size) for c in st.peaks_continuous])
        out["scene"][f] = st.smoothed_scene_calmness
        out["tuning"][f] = st.smoothed_tuning_grid_inaccuracy
    return out


@pytest.mark.parametrize("bpo,octaves,n_streams,n_frames,mode", [(36, 7, 256, 1000, "default"), (84, 7, 8, 300, "default"),
                                                                  (36, 5, 16, 300, "none"), (36, 7, 16, 300, "retuned")])
def test_gpu_streams_follow_the_host_state(bpo, octaves, n_streams, n_frames, mode):
    import os
    rng_ = P.VqtRange(55.0, octaves, bpo)
    nb = octaves * bpo
    words, max_peaks = (nb + 31) // 32, 64
    dt = 256.0 / 48000.0 * 3   # 16 ms frames
    x = _frames(n_streams, n_frames, nb, 1234 + bpo + n_streams)
    d_db = torch.from_numpy(x).cuda()
    b = P.AnalysisBatch(rng_, n_streams)
    if mode == "none":
        b.update_vqt_smoothing_duration(None)
    elif mode == "retuned":
        b.update_vqt_smoothing_duration(0.120)
    outs = {k: torch.zeros((n_streams, n_frames, nb), device="cuda") for k in FIELDS}
    outs["peak_mask"] = torch.zeros((n_streams, n_frames, words), dtype=torch.int32, device="cuda")
    outs["peak_count"] = torch.zeros((n_streams, n_frames), dtype=torch.int32, device="cuda")
    outs["center"] = torch.zeros((n_streams, n_frames, max_peaks), device="cuda")
    outs["size"] = torch.zeros((n_streams, n_frames, max_peaks), device="cuda")
    outs["scene_calmness"] = torch.zeros((n_streams, n_frames), device="cuda")
    outs["tuning_grid_inaccuracy"] = torch.zeros((n_streams, n_frames), device="cuda")
    # two calls: the state carries over (first 40 % of the frames, then the rest)
    cut = n_frames * 2 // 5
    first = {k: t[:, :cut].contiguous() for k, t in outs.items()}
    b.preprocess_device(d_db[:, :cut].contiguous(), cut, dt, first, max_peaks=max_peaks)
    rest = {k: t[:, cut:].contiguous() for k, t in outs.items()}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    b.preprocess_device(d_db[:, cut:].contiguous(), n_frames - cut, dt, rest, max_peaks=max_peaks)
    ev1.record()
    torch.cuda.synchronize()
    gpu_ms = ev0.elapsed_time(ev1)
    g = {k: torch.cat([first[k], rest[k]], dim=1).cpu().numpy() for k in outs}
    gmask = np.unpackbits(g["peak_mask"].view(np.uint8).reshape(n_streams, n_frames, -1), axis=-1, bitorder="little")[..., :nb].astype(bool)
    # host reference: every stream when few, else a spread of them (the host state takes ~50 us per frame through ctypes)
    check = list(range(n_streams)) if n_streams <= 16 else sorted(set(np.linspace(0, n_streams - 1, 24).astype(int)))
    import time
    t0 = time.perf_counter()
    exact = {k: [0, 0] for k in FIELDS + ("scene", "tuning", "center", "size")}
    for s in check:
        h = _host_reference(rng_, x[s], dt, mode)
        assert np.array_equal(gmask[s], h["mask"]), (s, np.argwhere(gmask[s] != h["mask"])[:5])
        assert np.array_equal(g["peak_count"][s], h["mask"].sum(axis=1))
        for k in FIELDS:
            a, w = g[k][s], h[k]
            exact[k][0] += int((a.view(np.uint32) == w.view(np.uint32)).sum()); exact[k][1] += a.size
            tol = dict(rtol=1e-5, atol=1e-5) if k in ("x_vqt_smoothed", "x_vqt_peakfiltered", "x_vqt_afterglow", "calmness") else dict(rtol=0, atol=2e-2)
            assert np.allclose(a, w, **tol), (s, k, np.abs(a - w).max())
        for k, gk in (("scene", "scene_calmness"), ("tuning", "tuning_grid_inaccuracy")):
            a, w = g[gk][s], h[k]
            exact[k][0] += int((a.view(np.uint32) == w.view(np.uint32)).sum()); exact[k][1] += a.size
            assert np.allclose(a, w, rtol=1e-5, atol=1e-5 if k == "scene" else 0.2), (s, k, np.abs(a - w).max())
        for f in range(n_frames):
            pc = h["pc"][f][:max_peaks]
            if pc:
                wc, ws = np.array(pc, np.float32).T
                ac, as_ = g["center"][s, f, :len(pc)], g["size"][s, f, :len(pc)]
                exact["center"][0] += int((ac.view(np.uint32) == wc.view(np.uint32)).sum()); exact["center"][1] += len(pc)
                exact["size"][0] += int((as_.view(np.uint32) == ws.view(np.uint32)).sum()); exact["size"][1] += len(pc)
                assert np.allclose(ac, wc, atol=3e-2) and np.allclose(as_, ws, atol=0.3), (s, f)
    host_s = time.perf_counter() - t0
    # final state through the getters
    for s in check[:4]:
        for k in FIELDS:
            assert np.array_equal(b.field(s, k), g[k][s, -1]), (s, k)
        sc, tu = b.scalars(s)
        assert sc == g["scene_calmness"][s, -1] and tu == g["tuning_grid_inaccuracy"][s, -1]
    # every stream (not only the compared ones) produced sane values
    assert np.isfinite(g["x_vqt_smoothed"]).all() and (g["scene_calmness"][:, -1] > 0).all()
    rate_gpu = n_streams * (n_frames - cut) / (gpu_ms * 1e-3)
    rate_host = len(check) * n_frames / host_s
    lines = [f"analysis batch {bpo} bpo x {octaves} oct, {n_streams} streams x {n_frames} frames, mode {mode}: GPU {rate_gpu / 1e6:.2f} M frames/s "
             f"({gpu_ms:.2f} ms for {n_frames - cut} frames of every stream), host AnalysisState through ctypes {rate_host / 1e3:.1f} k frames/s (one core)"]
    for k, (e, t) in exact.items():
        lines.append(f"    {k:20s} bit-identical {e}/{t} = {100.0 * e / max(t, 1):.4f} %")
    print("\n".join(lines))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/analysis_batch_r03.txt", "a") as fh:
        fh.write("\n".join(lines) + "\n")
    for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow"):   # the recurrence's own state: all but a handful of values equal bit for bit
        assert exact[k][0] >= 0.999 * exact[k][1], (k, exact[k])


def test_per_frame_durations():
    """frame_times_ns: a different frame_time per call of preprocess (analysis.rs:288 takes it per frame: a live consumer passes the
    time since its last frame) — jittered 8 ... 30 ms here; the recurrence state must follow the host object fed the same durations."""
    rng_ = P.VqtRange(55.0, 7, 36)
    nb, n_streams, n_frames = 252, 6, 240
    x = _frames(n_streams, n_frames, nb, 77)
    times = np.random.default_rng(5).uniform(0.008, 0.030, n_frames)
    times = np.round(times * 1e9) / 1e9   # whole nanoseconds: what crosses the ABI
    b = P.AnalysisBatch(rng_, n_streams)
    outs = {k: torch.zeros((n_streams, n_frames, nb), device="cuda") for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow")}
    outs["scene_calmness"] = torch.zeros((n_streams, n_frames), device="cuda")
    b.preprocess_device(torch.from_numpy(x).cuda(), n_frames, 0.0, outs, frame_times=list(times))
    torch.cuda.synchronize()
    g = {k: t.cpu().numpy() for k, t in outs.items()}
    same = total = 0
    for s in range(n_streams):
        st = P.AnalysisState.new(rng_)
        for f in range(n_frames):
            st.preprocess(x[s, f], float(times[f]))
            for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow"):
                a, w = g[k][s, f], np.asarray(getattr(st, k), np.float32)
                # the EMA weights are 1 - exp(-2 dt / horizon): the device rounds a double exp once, the host calls libm's expf, and with
                # thousands of distinct (dt, horizon) pairs the two differ by one ulp now and then (first at frame 34 here); the state
                # then follows within a few ulp
                assert np.allclose(a, w, rtol=2e-6, atol=2e-6), (s, f, k, np.abs(a - w).max())
                same += int((a.view(np.uint32) == w.view(np.uint32)).sum())
                total += a.size
            assert abs(g["scene_calmness"][s, f] - np.float32(st.smoothed_scene_calmness)) <= 2e-6, (s, f)
    assert same >= 0.98 * total, (same, total)


def test_batch_rejects_what_it_cannot_do():
    with pytest.raises(P.PvqError):
        P.AnalysisBatch(P.VqtRange(55.0, 7, 36), 4, device=-1)      # no CPU fallback
    with pytest.raises(P.PvqError):
        P.AnalysisBatch(P.VqtRange(55.0, 13, 84), 4)                 # 1 092 bins: unsupported
    b = P.AnalysisBatch(P.VqtRange(55.0, 2, 24), 3)
    with pytest.raises(P.PvqError):
        b.field(3, "calmness")
