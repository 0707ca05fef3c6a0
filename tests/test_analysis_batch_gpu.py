"""AnalysisState::preprocess for many streams on the GPU (pvq_analysis_batch_*, SURVEY.md 8f row 1's "one wave per stream" half)
pinned against the ORACLE (oracle/analysis_state.py — the NumPy-f32 restatement of analysis.rs:288-404, calmness.rs:23-95,
pitch_analysis.rs:12-75, afterglow.rs:10-36, util.rs:91-137 with glibc's expf / powf; its array form, proven bit-identical to the
literal scalar form by tests/test_oracle_analysis_vec.py), in the reference's DEFAULT smoothing mode (70 ms calmness-adaptive
EMA, analysis.rs:72-98) and two more, every pub field of every frame of every compared stream:

  * test_gpu_streams_follow_the_oracle      synthetic dB frames -> GPU batch vs the oracle (the product's host AnalysisState
                                            rides along as a second reference on a few streams);
  * test_pipeline_pcm_to_analysis_on_device PCM -> GPU VQT -> GPU batch with no host hop, against oracle C VQT -> oracle state
                                            (config-5 piano roll + white noise), on the block-DFT and the FFT path;
  * test_vqt_close_frequencies_batch        the reference's own end-to-end test of find_peaks (lib.rs:16-48: exactly 2 peaks)
                                            through GPU VQT + GPU batch, one stream per test tone;
  * test_per_frame_durations                a different frame_time per preprocess call.

What is compared how.  Peak index sets (bit masks) and counts: EQUAL in every frame on identical input frames.  Float fields: bit
for bit; the share of exactly equal values is reported, and what differs is bounded by the tolerances written at TOL below — they
are what the bit-identity shares justify (values that differ do so by a libm call rounded the other way), not the loose bounds
the round-3 version of this file carried."""
import os
import time

import numpy as np
import pytest

import oracle as O
from oracle.analysis_state import OracleAnalysisStateVec
import pitchvis_amd as P
from helpers import get_geom, mask_to_indices, report, white_noise
from synth import piano_roll

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

OUT = "analysis_batch_r05.txt"
FIELDS = ("x_vqt_smoothed", "x_vqt_peakfiltered", "x_vqt_afterglow", "calmness", "pitch_accuracy", "pitch_deviation")
# GPU batch vs oracle on IDENTICAL dB frames.  The recurrence state (EMAs of dB values 0..60 and of calmness 0..1) may differ by
# an ulp where an EMA weight rounded the other way; the continuous peaks come out of an ill-conditioned f32 parabola in
# ln-frequency (peak_detection.rs:91-118) that turns one ulp of a smoothed value into up to ~1e-4 bin; sizes are dB values
# interpolated at that centre; pitch deviation = centre * 12 / bpo - round(.), accuracy = 1 - 2 |deviation|; the tuning
# inaccuracy is 100 x a power-weighted mean of |deviation|.
# Measured (profiles/r04_analysis_batch.txt): state fields and peak sets identical in every value; centre <= 1.6e-5 bin, size <= 1.5e-4 dB,
# deviation <= 8e-6, accuracy <= 1.6e-5, tuning <= 6e-6 — the bounds below leave a factor ~3.
TOL = dict(x_vqt_smoothed=1e-5, x_vqt_peakfiltered=1e-5, x_vqt_afterglow=1e-5, calmness=2e-6, scene=2e-6,
           center=5e-5, size=5e-4, pitch_deviation=2e-5, pitch_accuracy=4e-5, tuning=5e-5)


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


def _set_mode(obj, mode, ns=False):
    if mode == "none":
        obj.update_vqt_smoothing_duration(None)
    elif mode == "retuned":
        obj.update_vqt_smoothing_duration(120_000_000 if ns else 0.120)


def _oracle_reference(rng_, x, times_ns, mode, okw=None):
    """oracle/analysis_state.py over one stream: every pub field per frame (times_ns: one duration, or one per frame)"""
    st = OracleAnalysisStateVec(rng_.min_freq, rng_.octaves, rng_.buckets_per_octave, **(okw or {}))
    _set_mode(st, mode, ns=True)
    nf, nb = x.shape
    out = {k: np.empty((nf, nb), np.float32) for k in FIELDS}
    out["mask"] = np.zeros((nf, nb), bool)
    out["scene"] = np.empty(nf, np.float32); out["tuning"] = np.empty(nf, np.float32)
    out["pc"] = []
    for f in range(nf):
        st.preprocess(x[f], int(times_ns[f]) if np.ndim(times_ns) else int(times_ns))
        out["x_vqt_smoothed"][f] = st.sm; out["x_vqt_peakfiltered"][f] = st.peakfiltered; out["x_vqt_afterglow"][f] = st.afterglow
        out["calmness"][f] = st.calm; out["pitch_accuracy"][f] = st.pitch_accuracy; out["pitch_deviation"][f] = st.pitch_deviation
        out["mask"][f, st.peaks] = True
        out["pc"].append((st.centers.copy(), st.sizes.copy()))
        out["scene"][f] = st.scene; out["tuning"][f] = st.tuning
    return out


def _host_reference(rng_, x, dt, mode, fp=None):
    """the product's host AnalysisState over one stream (second reference)"""
    st = P.AnalysisState(rng_, fp) if fp is not None else P.AnalysisState.new(rng_)
    _set_mode(st, mode)
    nf, nb = x.shape
    out = {k: np.empty((nf, nb), np.float32) for k in FIELDS}
    out["mask"] = np.zeros((nf, nb), bool)
    out["scene"] = np.empty(nf, np.float32); out["tuning"] = np.empty(nf, np.float32)
    for f in range(nf):
        st.preprocess(x[f], dt)
        for k in FIELDS:
            out[k][f] = getattr(st, k)
        for p in st.peaks:
            out["mask"][f, p] = True
        out["scene"][f] = st.smoothed_scene_calmness
        out["tuning"][f] = st.smoothed_tuning_grid_inaccuracy
    return out


def _alloc_outputs(n_streams, n_frames, nb, max_peaks):
    words = (nb + 31) // 32
    outs = {k: torch.zeros((n_streams, n_frames, nb), device="cuda") for k in FIELDS}
    outs["peak_mask"] = torch.zeros((n_streams, n_frames, words), dtype=torch.int32, device="cuda")
    outs["peak_count"] = torch.zeros((n_streams, n_frames), dtype=torch.int32, device="cuda")
    outs["center"] = torch.zeros((n_streams, n_frames, max_peaks), device="cuda")
    outs["size"] = torch.zeros((n_streams, n_frames, max_peaks), device="cuda")
    outs["scene_calmness"] = torch.zeros((n_streams, n_frames), device="cuda")
    outs["tuning_grid_inaccuracy"] = torch.zeros((n_streams, n_frames), device="cuda")
    return outs


def _to_host(outs, nb):
    g = {k: t.cpu().numpy() for k, t in outs.items()}
    ns, nf = g["peak_count"].shape
    g["mask"] = np.unpackbits(g["peak_mask"].view(np.uint8).reshape(ns, nf, -1), axis=-1, bitorder="little")[..., :nb].astype(bool)
    return g


class _Tally:
    """bit-identity shares and the largest difference per field"""

    def __init__(self):
        self.same, self.total, self.worst = {}, {}, {}

    def add(self, key, a, w):
        a, w = np.asarray(a, np.float32), np.asarray(w, np.float32)
        self.same[key] = self.same.get(key, 0) + int((a.view(np.uint32) == w.view(np.uint32)).sum())
        self.total[key] = self.total.get(key, 0) + a.size
        if a.size:
            self.worst[key] = max(self.worst.get(key, 0.0), float(np.abs(a.astype(np.float64) - w).max()))

    def lines(self):
        return [f"    {k:20s} bit-identical {self.same[k]}/{self.total[k]} = {100.0 * self.same[k] / max(self.total[k], 1):.4f} %, max |diff| {self.worst.get(k, 0.0):.3e}"
                for k in self.total]


def _compare_stream(tally, g, s, w, n_frames, max_peaks, tol, where):
    """every pub field of every frame of stream s: GPU (g) against a reference (w); identical input frames -> equal peak sets"""
    assert np.array_equal(g["mask"][s], w["mask"]), (where, s, np.argwhere(g["mask"][s] != w["mask"])[:5])
    assert np.array_equal(g["peak_count"][s], w["mask"].sum(axis=1)), (where, s)
    for k in FIELDS:
        tally.add(k, g[k][s], w[k])
        assert np.abs(g[k][s] - w[k]).max() <= tol[k], (where, s, k, float(np.abs(g[k][s] - w[k]).max()))
    tally.add("scene", g["scene_calmness"][s], w["scene"])
    tally.add("tuning", g["tuning_grid_inaccuracy"][s], w["tuning"])
    assert np.abs(g["scene_calmness"][s] - w["scene"]).max() <= tol["scene"], (where, s)
    assert np.abs(g["tuning_grid_inaccuracy"][s] - w["tuning"]).max() <= tol["tuning"], (where, s, float(np.abs(g["tuning_grid_inaccuracy"][s] - w["tuning"]).max()))
    if "pc" in w:
        for f in range(n_frames):
            wc, ws = w["pc"][f]
            k = min(wc.size, max_peaks)
            if k:
                ac, as_ = g["center"][s, f, :k], g["size"][s, f, :k]
                tally.add("center", ac, wc[:k]); tally.add("size", as_, ws[:k])
                assert np.abs(ac - wc[:k]).max() <= tol["center"] and np.abs(as_ - ws[:k]).max() <= tol["size"], \
                    (where, s, f, float(np.abs(ac - wc[:k]).max()), float(np.abs(as_ - ws[:k]).max()))


@pytest.mark.parametrize("bpo,octaves,n_streams,n_frames,mode", [(36, 7, 256, 1000, "default"), (84, 7, 8, 300, "default"),
                                                                  (36, 5, 16, 300, "none"), (36, 7, 16, 300, "retuned"),
                                                                  (36, 10, 6, 260, "default"), (84, 9, 5, 240, "default"), (84, 12, 4, 220, "default")])   # (360 / 756 / 1 008 bins: 6 / 12 / 16 bins per lane)
def test_gpu_streams_follow_the_oracle(bpo, octaves, n_streams, n_frames, mode):
    rng_ = P.VqtRange(55.0, octaves, bpo)
    nb = octaves * bpo
    max_peaks = 64
    dt = 256.0 / 48000.0 * 3   # 16 ms frames
    dt_ns = int(round(dt * 1e9))
    x = _frames(n_streams, n_frames, nb, 1234 + bpo + n_streams)
    d_db = torch.from_numpy(x).cuda()
    b = P.AnalysisBatch(rng_, n_streams)
    _set_mode(b, mode)
    outs = _alloc_outputs(n_streams, n_frames, nb, max_peaks)
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
    g = _to_host({k: torch.cat([first[k], rest[k]], dim=1) for k in outs}, nb)
    # the oracle over every stream when few, else over a spread of 24 of them (0.2 ms per frame)
    check = list(range(n_streams)) if n_streams <= 16 else sorted(set(np.linspace(0, n_streams - 1, 24).astype(int)))
    t0 = time.perf_counter()
    tally = _Tally()
    for s in check:
        _compare_stream(tally, g, s, _oracle_reference(rng_, x[s], dt_ns, mode), n_frames, max_peaks, TOL, "oracle")
    oracle_s = time.perf_counter() - t0
    # second reference: the product's host AnalysisState on a few streams (it calls the same glibc routines as the oracle)
    host_tally = _Tally()
    for s in check[:3]:
        _compare_stream(host_tally, g, s, _host_reference(rng_, x[s], dt, mode), n_frames, max_peaks, TOL, "host")
    # final state through the getters
    for s in check[:4]:
        for k in FIELDS:
            assert np.array_equal(b.field(s, k), g[k][s, -1]), (s, k)
        sc, tu = b.scalars(s)
        assert sc == g["scene_calmness"][s, -1] and tu == g["tuning_grid_inaccuracy"][s, -1]
    # every stream (not only the compared ones) produced sane values
    assert np.isfinite(g["x_vqt_smoothed"]).all() and (g["scene_calmness"][:, -1] > 0).all()
    rate_gpu = n_streams * (n_frames - cut) / (gpu_ms * 1e-3)
    lines = [f"analysis batch vs ORACLE, {bpo} bpo x {octaves} oct, {n_streams} streams x {n_frames} frames, mode {mode}: GPU {rate_gpu / 1e6:.2f} M frames/s "
             f"({gpu_ms:.2f} ms for {n_frames - cut} frames of every stream); oracle (array form, one core) {len(check) * n_frames / oracle_s / 1e3:.1f} k frames/s; "
             f"{len(check)} streams compared, peak sets equal in all {len(check) * n_frames} frames"]
    lines += tally.lines()
    lines.append("  against the product's host AnalysisState (3 streams):")
    lines += host_tally.lines()
    for ln in lines:
        report(OUT, ln)
    for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow"):   # the recurrence's own state: all but a handful of values equal bit for bit
        assert tally.same[k] >= 0.999 * tally.total[k], (k, tally.same[k], tally.total[k])
    assert tally.same["scene"] >= 0.99 * tally.total["scene"]   # (a power-weighted mean: one powf rounded the other way moves it by an ulp)


# every field of AnalysisParameters (analysis.rs:36-65) moved off its default (analysis.rs:72-98): as the product's parameter object and as
# the oracle's keywords
NONDEFAULT = dict(
    full=dict(peak_config=(8.0, 3.0), bassline_peak_config=(4.0, 2.5), highest_bassnote=20, vqt_smoothing_duration_base=0.040,
              vqt_smoothing_calmness_min=0.5, vqt_smoothing_calmness_max=3.0, note_calmness_smoothing_duration=2.0,
              scene_calmness_smoothing_duration=0.5, tuning_inaccuracy_smoothing_duration=3.0, harmonic_threshold=0.2),
    oracle=dict(peak=(8.0, 3.0), bass=(4.0, 2.5), highest_bassnote=20, base_ns=40_000_000, cmin=0.5, cmax=3.0, note_ns=2_000_000_000,
                scene_ns=500_000_000, tuning_ns=3_000_000_000, harmonic_threshold=0.2))


def _nondefault_params():
    f = NONDEFAULT["full"]
    return P.FullAnalysisParameters(peak_config=P.PeakDetectionParameters(*f["peak_config"]), bassline_peak_config=P.PeakDetectionParameters(*f["bassline_peak_config"]),
                                    **{k: v for k, v in f.items() if k not in ("peak_config", "bassline_peak_config")})


@pytest.mark.parametrize("bpo,octaves,n_streams,n_frames,per_frame_dt", [(36, 7, 12, 400, False), (84, 7, 6, 260, False), (36, 7, 6, 300, True)])
def test_gpu_streams_non_default_parameters(bpo, octaves, n_streams, n_frames, per_frame_dt):
    """The GPU batch with EVERY field of AnalysisParameters off its default (analysis.rs:36-98: peak 8.0 / 3.0, bass 4.0 / 2.5 with the split
    at bin 20, a 40 ms base horizon stretched by calmness 0.5 ... 3.0, note / scene / tuning horizons 2 s / 0.5 s / 3 s, harmonic threshold
    0.2) against the oracle with the same values: the host-built EMA-weight table is sized from base * 1.5 * calmness_max + 3 entries, the
    bass / general split and both find_peaks configurations move, the promotion test changes.  The third case gives every frame its own
    duration (one table row per distinct frame time)."""
    rng_ = P.VqtRange(55.0, octaves, bpo)
    nb = octaves * bpo
    max_peaks = 64
    dt = 0.016
    rng = np.random.default_rng(99 + bpo)
    times = (np.round(rng.uniform(0.004, 0.034, n_frames) * 1e6) * 1e-6) if per_frame_dt else None
    times_ns = np.round(times * 1e9).astype(np.int64) if per_frame_dt else int(round(dt * 1e9))
    x = _frames(n_streams, n_frames, nb, 4321 + bpo + n_streams)
    fp = _nondefault_params()
    b = P.AnalysisBatch(rng_, n_streams, params=fp)
    outs = _alloc_outputs(n_streams, n_frames, nb, max_peaks)
    b.preprocess_device(torch.from_numpy(x).cuda(), n_frames, dt, outs, max_peaks=max_peaks, frame_times=times)
    torch.cuda.synchronize()
    g = _to_host(outs, nb)
    tally, host_tally = _Tally(), _Tally()
    default_masks_differ = 0
    for s in range(n_streams):
        w = _oracle_reference(rng_, x[s], times_ns, "default", NONDEFAULT["oracle"])
        _compare_stream(tally, g, s, w, n_frames, max_peaks, TOL, "oracle, non-default parameters")
        if s < 2:
            default_masks_differ += int((_oracle_reference(rng_, x[s], times_ns, "default")["mask"] != w["mask"]).sum())
            if not per_frame_dt:
                _compare_stream(host_tally, g, s, _host_reference(rng_, x[s], dt, "default", fp), n_frames, max_peaks, TOL, "host, non-default parameters")
    assert default_masks_differ > 0   # (the moved parameters do change the answer: this is not the default case again)
    lines = [f"analysis batch vs ORACLE with non-default AnalysisParameters, {bpo} bpo x {octaves} oct, {n_streams} streams x {n_frames} frames"
             f"{', a duration per frame' if per_frame_dt else ''}: peak sets equal in all {n_streams * n_frames} frames ({default_masks_differ} mask bits differ from the default parameters' on 2 streams)"]
    lines += tally.lines()
    if host_tally.total:
        lines.append("  against the product's host AnalysisState (2 streams):")
        lines += host_tally.lines()
    for ln in lines:
        report(OUT, ln)
    for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow"):
        assert tally.same[k] >= 0.999 * tally.total[k], (k, tally.same[k], tally.total[k])


def _margin(frame, b, rng_, ap):
    """distance (dB) of bin b of a smoothed frame from the threshold that decides whether it is a peak"""
    from test_parity_evidence_gpu import _decisive_margin
    return _decisive_margin(frame, b, rng_.buckets_per_octave, ap)


@pytest.mark.parametrize("hop,algo", [(1024, P.ALGO_BLOCKDFT), (800, P.ALGO_FFT)])
def test_pipeline_pcm_to_analysis_on_device(hop, algo):
    """The reference's default pipeline end to end on the device: PCM -> Vqt::calculate_vqt_instant_in_db per hop -> AnalysisState::
    preprocess with the default calmness-adaptive smoothing (vqt.rs:866-916 -> analysis.rs:288-404), N streams, no host hop between
    the two stages — against oracle C VQT -> oracle analysis state on the same PCM.  hop 1024 = 21.3 ms frames on the block-DFT
    path; hop 800 = the viewer's 60 fps cadence at 48 kHz (pitchvis_viewer/src/app/desktop_app.rs:18) on the FFT path.  The two dB
    streams differ by the dB parity tolerance (<= 1e-2 dB, 2e-4 on strong bins), so: peak index sets equal except in frames where a
    smoothed value sits within that tolerance of the deciding threshold (each listed with its margin); the recurrence state within
    the dB parity bar.  Continuous peaks: on IDENTICAL frames the GPU equals the oracle to 1.6e-5 bin (part (a) below, TOL); between
    the two chains the bound is what the reference's own arithmetic allows — its f32 log-frequency parabola (peak_detection.rs:91-118)
    turns a 1e-5 dB change of its input into up to 1.5e-2 bin and 0.13 dB (median 2e-3 bin; measured on the ORACLE alone by
    perturbing its input, see test_parabola_conditioning_of_the_reference below) — so: centre <= 3e-2 bin, size <= 0.3 dB, tuning
    inaccuracy <= 5e-2 cent, the bounds tests/test_analysis_state.py uses between the host state and the oracle for the same reason."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    v.set_algo(algo)
    ov = O.OracleVqt(op)
    ap = O.OracleAnalysisParams()
    n_streams, seconds = 8, 6.0
    nf = int(op.sr * seconds) // hop
    dt = hop / op.sr
    dt_ns = int(round(dt * 1e9))
    nb, max_peaks = v.n_bins, 64
    pcm = []
    for s in range(n_streams):
        x, _ = piano_roll(op.sr, seconds, 100 + s)
        pcm.append((x + white_noise(x.size, 200 + s, amp=0.01))[:nf * hop].astype(np.float32))
    # stage 1 on the device: ONE many-streams call writes every stream's dB frames into the [stream][frame][bin] tensor stage 2 reads
    d_db = torch.empty((n_streams, nf, nb), device="cuda")
    d_pcm = [torch.from_numpy(x).cuda() for x in pcm]
    v.batch_streams_device(d_pcm, hop, [nf] * n_streams, d_db, nf)
    assert v.last_algo() == algo
    # stage 2 on the device, straight from d_db
    b = P.AnalysisBatch(pp.range, n_streams)
    outs = _alloc_outputs(n_streams, nf, nb, max_peaks)
    b.preprocess_device(d_db, nf, dt, outs, max_peaks=max_peaks)
    torch.cuda.synchronize()
    v.input_status()
    g = _to_host(outs, nb)
    gdb = d_db.cpu().numpy()
    worst = dict(db=0.0, smoothed=0.0, afterglow=0.0, calm=0.0, scene=0.0, center=0.0, size=0.0, tuning=0.0)
    n_peaks = n_diff_frames = n_frames_total = 0
    for s in range(n_streams):
        wdb = ov.calculate_batch(pcm[s], hop, nf)
        worst["db"] = max(worst["db"], float(np.abs(gdb[s] - wdb).max()))
        assert np.abs(gdb[s] - wdb).max() <= 1e-2
        w = _oracle_reference(pp.range, wdb, dt_ns, "default")
        # (a) the batch kernel itself on the GPU's own frames: exact peak sets, state to the ulp-level tolerances
        _compare_stream(_Tally(), g, s, _oracle_reference(pp.range, gdb[s], dt_ns, "default"), nf, max_peaks, TOL, "oracle-on-gpu-frames")
        # (b) the whole chain against the all-CPU chain
        for k, key in (("x_vqt_smoothed", "smoothed"), ("x_vqt_afterglow", "afterglow"), ("calmness", "calm")):
            worst[key] = max(worst[key], float(np.abs(g[k][s] - w[k]).max()))
        worst["scene"] = max(worst["scene"], float(np.abs(g["scene_calmness"][s] - w["scene"]).max()))
        worst["tuning"] = max(worst["tuning"], float(np.abs(g["tuning_grid_inaccuracy"][s] - w["tuning"]).max()))
        for f in range(nf):
            n_frames_total += 1
            gp, wp = np.nonzero(g["mask"][s, f])[0], np.nonzero(w["mask"][f])[0]
            n_peaks += wp.size
            if np.array_equal(gp, wp):
                wc, ws = w["pc"][f]
                k = min(wc.size, max_peaks)
                if k:
                    worst["center"] = max(worst["center"], float(np.abs(g["center"][s, f, :k] - wc[:k]).max()))
                    worst["size"] = max(worst["size"], float(np.abs(g["size"][s, f, :k] - ws[:k]).max()))
            else:
                n_diff_frames += 1
                for bin_ in sorted(set(gp.tolist()) ^ set(wp.tolist())):
                    m = _margin(w["x_vqt_smoothed"][f], int(bin_), pp.range, ap)
                    report(OUT, f"pipeline peakdiff [hop {hop}] stream {s} frame {f} bin {bin_}: smoothed dB gpu {g['x_vqt_smoothed'][s, f, bin_]:.5f} "
                                f"oracle {w['x_vqt_smoothed'][f, bin_]:.5f}; margin to the deciding threshold {m:.2e} dB")
                    assert m <= 2e-2, (s, f, bin_, m)
    report(OUT, f"# pipeline PCM -> GPU VQT -> GPU AnalysisBatch vs oracle VQT -> oracle state, hop {hop} ({'block-DFT' if algo == P.ALGO_BLOCKDFT else 'FFT'} path), "
                f"{n_streams} streams x {nf} frames, {n_peaks} oracle peaks: {n_diff_frames} frames with a different peak set; max |diff|: "
                + ", ".join(f"{k} {v_:.2e}" for k, v_ in worst.items()))
    assert n_diff_frames <= max(1, n_frames_total // 100)
    # the recurrence state follows the dB frames: within the dB parity bar (an EMA is a convex combination of its inputs)
    assert worst["smoothed"] <= 1e-2 and worst["afterglow"] <= 1e-2
    # calmness only moves with the raw frame's peak set: equal sets -> ulp-level agreement; a straddling frame moves it by one EMA step
    assert worst["calm"] <= 2e-2 and worst["scene"] <= 2e-2
    assert worst["center"] <= 3e-2 and worst["size"] <= 0.3 and worst["tuning"] <= 5e-2, worst


def test_parabola_conditioning_of_the_reference():
    """Why centre / size of two chains whose dB frames differ by 1e-5 cannot be compared more tightly than ~1e-2 bin: the reference's
    enhance_peaks_continuous (peak_detection.rs:91-118) fits a parabola through three (ln f, dB) points in f32; the numerator
    l2 (a1 - a0) + l0 (a2 - a1) + l1 (a0 - a2) cancels to a few percent of its terms.  Measured here on the ORACLE alone (no GPU
    value involved): the same frames, once as they are and once with every bin moved by a uniform random amount below 1e-5 dB."""
    op = O.OracleParams(sr=48000.0, octaves=7, buckets_per_octave=36)
    ov = O.OracleVqt(op)
    x, _ = piano_roll(op.sr, 3.0, 100)
    pcm = (x + white_noise(x.size, 200, amp=0.01)).astype(np.float32)
    db = ov.calculate_batch(pcm, 1024, pcm.size // 1024)
    rng = np.random.default_rng(0)
    dc, ds = [], []
    for f in range(db.shape[0]):
        i0, c0, s0 = O.analyze_frame(db[f], 55.0, 7, 36)
        i1, c1, s1 = O.analyze_frame((db[f] + rng.uniform(-1e-5, 1e-5, db[f].size).astype(np.float32)).astype(np.float32), 55.0, 7, 36)
        if np.array_equal(i0, i1) and c0.size:
            dc += np.abs(c0 - c1).tolist(); ds += np.abs(s0 - s1).tolist()
    dc, ds = np.array(dc), np.array(ds)
    report(OUT, f"# conditioning of the reference's f32 parabola (oracle vs oracle, input moved by < 1e-5 dB, {dc.size} peaks): centre change median "
                f"{np.median(dc):.2e}, 99 % {np.quantile(dc, 0.99):.2e}, max {dc.max():.2e} bin; size change max {ds.max():.2e} dB")
    assert 1e-3 <= dc.max() <= 3e-2 and ds.max() <= 0.3    # (the lower bound: if this ever became well-conditioned the pipeline bounds above should tighten)


def test_vqt_close_frequencies_batch():
    """pitchvis_analysis/src/lib.rs:16-48 (`test_vqt_close_frequencies`) through the device pipeline: two sines a semitone apart, from
    2.6 octaves above min_freq to half an octave below the top in 1/30-octave steps, default VqtParameters (22 050 Hz, 7 x 84 bins);
    each tone pair is its own STREAM of the batch: a fresh AnalysisState fed one frame with frame_time = 1100 ms ->
    analysis.peaks.len() == 2 for every one of them — the reference's only test that pins find_peaks end to end."""
    p = O.default_params()
    pp = P.VqtParameters.default()
    v = P.Vqt.new(pp, 0)
    sub = 30
    cases = list(range(int(np.float32(2.6) * np.float32(sub)), p.octaves * sub - sub // 2))
    sounds = []
    for i in cases:
        ln = np.float32(i) / np.float32(sub)
        f1 = np.float32(p.min_freq) * np.float32(2.0) ** ln
        f2 = np.float32(p.min_freq) * np.float32(2.0) ** (ln + np.float32(1.0 / 12.0))
        sounds.append(O.test_create_sines(p, [f1, f2]))
    n = len(cases)
    # Vqt::calculate_vqt_instant_in_db of every sound in one call: hop = n_fft, frame i sees exactly sound i
    d_pcm = torch.from_numpy(np.concatenate(sounds)).cuda()
    d_db = torch.empty((n, 1, v.n_bins), device="cuda")        # [stream][frame = 1][bin]
    v.calculate_batch_db_device(d_pcm, p.n_fft, n, d_db)
    b = P.AnalysisBatch(pp.range, n)
    outs = {"peak_count": torch.zeros((n, 1), dtype=torch.int32, device="cuda"),
            "peak_mask": torch.zeros((n, 1, (v.n_bins + 31) // 32), dtype=torch.int32, device="cuda")}
    b.preprocess_device(d_db, 1, 1.100, outs)
    torch.cuda.synchronize()
    counts = outs["peak_count"].cpu().numpy()[:, 0]
    report(OUT, f"# test_vqt_close_frequencies on the device (GPU VQT -> GPU AnalysisBatch, {n} tone pairs = {n} streams x 1 frame): peak counts {sorted(set(counts.tolist()))}")
    assert (counts == 2).all(), [(cases[i], int(c)) for i, c in enumerate(counts) if c != 2]
    # and the two peaks are where the oracle's chain puts them
    ov = O.OracleVqt(p)
    mask = outs["peak_mask"].cpu().numpy().view(np.uint32)
    for i in range(0, n, 9):
        st = OracleAnalysisStateVec(p.min_freq, p.octaves, p.buckets_per_octave)
        st.preprocess(ov.calculate_vqt_instant_in_db(sounds[i]), 1_100_000_000)
        assert np.array_equal(mask_to_indices(mask[i, 0], v.n_bins), np.sort(st.peaks)), cases[i]


def test_per_frame_durations():
    """frame_times_ns: a different frame_time per call of preprocess (analysis.rs:288 takes it per frame: a live consumer passes the
    time since its last frame) — jittered 8 ... 30 ms here; every pub field against the oracle fed the same durations."""
    rng_ = P.VqtRange(55.0, 7, 36)
    nb, n_streams, n_frames, max_peaks = 252, 6, 240, 64
    x = _frames(n_streams, n_frames, nb, 77)
    times = np.random.default_rng(5).uniform(0.008, 0.030, n_frames)
    times_ns = np.round(times * 1e9).astype(np.int64)   # whole nanoseconds: what crosses the ABI
    b = P.AnalysisBatch(rng_, n_streams)
    outs = _alloc_outputs(n_streams, n_frames, nb, max_peaks)
    b.preprocess_device(torch.from_numpy(x).cuda(), n_frames, 0.0, outs, max_peaks=max_peaks, frame_times=list(times_ns / 1e9))
    torch.cuda.synchronize()
    g = _to_host(outs, nb)
    tally = _Tally()
    for s in range(n_streams):
        _compare_stream(tally, g, s, _oracle_reference(rng_, x[s], times_ns, "default"), n_frames, max_peaks, TOL, "oracle")
    report(OUT, f"analysis batch vs ORACLE with per-frame durations (8 ... 30 ms), {n_streams} streams x {n_frames} frames:")
    for ln in tally.lines():
        report(OUT, ln)
    for k in ("x_vqt_smoothed", "calmness", "x_vqt_afterglow", "scene"):
        assert tally.same[k] >= 0.98 * tally.total[k], (k, tally.same[k], tally.total[k])


def test_batch_rejects_what_it_cannot_do():
    with pytest.raises(P.PvqError):
        P.AnalysisBatch(P.VqtRange(55.0, 7, 36), 4, device=-1)      # no CPU fallback
    with pytest.raises(P.PvqError):
        P.AnalysisBatch(P.VqtRange(55.0, 13, 84), 4)                 # 1 092 bins: unsupported
    b = P.AnalysisBatch(P.VqtRange(55.0, 2, 24), 3)
    with pytest.raises(P.PvqError):
        b.field(3, "calmness")
