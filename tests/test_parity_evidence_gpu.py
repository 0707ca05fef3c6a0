"""Measured error of the HIP path, written down instead of bounded loosely ("What's weak" 2 and 3).

1. Error versus level: for every bin of every frame, the error against the exact-in-f64 transform of the
   same kernel (oracle/model_f64.py), binned by how far the bin sits below its frame's strongest bin, for
   the GPU (both algorithms and both arithmetics) and for the f32 CPU oracle side by side.  What an f32
   evaluation of sum_n x[n] g_k[n] can promise is an ABSOLUTE error proportional to the input it sums
   (eps * sqrt(sr) * max|x| * a modest growth factor), not a relative error per bin: a bin 60 dB down
   carries the same absolute error as the strongest one, so its relative error is 1000 x larger — on the
   CPU path exactly as on the GPU.  The table shows that; the assertions are on the absolute error:
   <= 4e-7 of the input scale sqrt(sr) max|x| for every bin of every frame (the universal bound; measured
   2.3e-7 at worst, on coherent sweeps, where the CPU path itself reaches 1.9e-7), <= 1e-5 of the frame
   maximum (north_star's figure) at every level for every frame whose maximum reaches 1 % of the input scale, <= 2e-6 for frames that reach 10 %, per-bin relative <= 1e-5 within 10 dB of
   the maximum, and the GPU within 4 x the oracle's own error + 2e-7.
2. Stream start: the first frames after silence see the signal only through the Hann tails, their
   coefficients are the residue of large cancelling terms, and errors relative to the (tiny) frame maximum
   reach 1e-3.  The table prints, per frame, the frame maximum and both errors against the input scale
   sqrt(sr) * max|x|: the error on that scale is the same 1e-7 as everywhere else.
3. End-to-end peak sets: every frame whose GPU peak set differs from the oracle's is listed with the bin
   and its margin to the threshold that decides it (height or prominence); a margin larger than the dB
   parity tolerance fails the test.
Output: gpurun_out/parity_evidence_r02.txt (the builder copies it to profiles/).
"""
import numpy as np
import pytest
from scipy.signal import peak_prominences

import oracle as O
import pitchvis_amd as P
from oracle import model_f64 as MF
from helpers import get_geom, mask_to_indices, report, sine_sweep, three_regime, white_noise
from test_parity_gpu import input_peak, run_gpu, _set_algo

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

OUT = "parity_evidence_r02.txt"
ALGOS = [P.ALGO_FFT, P.ALGO_BLOCKDFT, "blockdft_bf16x3"]
NAME = {P.ALGO_FFT: "fft", P.ALGO_BLOCKDFT: "blockdft", "blockdft_bf16x3": "blockdft_bf16x3"}
EDGES = np.arange(0, 130, 10)   # dB below the frame's strongest bin


def _cases(op):
    hop = 256
    yield "noise", white_noise(33000 + hop * 72, 0x5EED0001), hop, 72, 33000
    yield "sweep", sine_sweep(hop * 375, op.sr), hop, 375, 0
    yield "three_regime", three_regime(hop * 384, op.sr, 3), hop, 384, 0


def _level_table(tag, cx, wcx, truth, scale):
    """rows: level bin -> count, GPU max / median relative error, oracle max / median relative error, both max absolute
    errors in units of `scale` (per frame)"""
    at = np.abs(truth)
    fmax = at.max(axis=1, keepdims=True)
    live = (fmax[:, 0] > 0)
    lvl = np.full(at.shape, np.inf)
    np.divide(at, fmax, out=lvl, where=fmax > 0)
    with np.errstate(divide="ignore"):
        lvl = -20 * np.log10(lvl)
    eg = np.abs(cx - truth)
    ec = np.abs(wcx - truth)
    rows = []
    report(OUT, f"# {tag}: level below the frame's strongest bin -> bins, GPU rel err max / median, oracle rel err max / median, "
                f"GPU abs err / scale max, oracle abs err / scale max")
    for lo, hi in zip(EDGES[:-1], EDGES[1:]):
        m = (lvl >= lo) & (lvl < hi) & live[:, None]
        if not m.any():
            continue
        rg = eg[m] / at[m]
        rc = ec[m] / at[m]
        ag = (eg / scale)[m]
        ac = (ec / scale)[m]
        rows.append((lo, hi, int(m.sum()), rg.max(), np.median(rg), rc.max(), np.median(rc), ag.max(), ac.max()))
        report(OUT, f"{tag} {lo:3d}..{hi:3d} dB  n={int(m.sum()):7d}  gpu {rg.max():.2e} / {np.median(rg):.2e}   "
                    f"oracle {rc.max():.2e} / {np.median(rc):.2e}   abs/scale gpu {ag.max():.2e} oracle {ac.max():.2e}")
    return rows


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("geom", ["bench_48k_252", "default_22k_588"])
def test_error_versus_level(geom, algo):
    pp, op = get_geom(geom)
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    model = MF.from_oracle_params(op, values_from=ov)
    _set_algo(v, algo)
    for case, pcm, hop, nf, n_lead in _cases(op):
        _, cx = run_gpu(v, pcm, hop, nf, n_lead)
        assert v.last_algo() == (P.ALGO_FFT if algo == P.ALGO_FFT else P.ALGO_BLOCKDFT)
        _, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
        truth = model.batch_complex(pcm, hop, nf, n_lead=n_lead)
        xp = input_peak(pcm, hop, nf, n_lead, v.window_union)
        inp = np.sqrt(op.sr) * xp                                  # what a full-scale sine of the window's peak amplitude would give
        fmax = np.abs(truth).max(axis=1)
        well = (fmax >= 0.01 * inp) & (fmax > 0)                   # frames whose output is not the residue of cancelling terms
        live = xp > 0
        # the universal bound: absolute error against the input scale, every bin of every frame
        ei_g = (np.abs(cx - truth).max(axis=1)[live] / inp[live]).max()
        ei_c = (np.abs(wcx - truth).max(axis=1)[live] / inp[live]).max()
        report(OUT, f"{geom} {NAME[algo]} {case}: {int(live.sum())} frames, max abs err / (sqrt(sr) max|x|): gpu {ei_g:.2e}, oracle {ei_c:.2e}")
        assert ei_g <= 4e-7 and ei_g <= 2.0 * ei_c + 1e-7, (geom, algo, case, ei_g, ei_c)   # measured: <= 2.3e-7 (coherent sweeps), the CPU path <= 1.9e-7
        if well.any():
            scale = fmax[well][:, None]
            rows = _level_table(f"{geom} {NAME[algo]} {case}", cx[well], wcx[well], truth[well], scale)
            for (lo, hi, n, rgm, rgmed, rcm, rcmed, agm, acm) in rows:
                assert agm <= 1e-5, (geom, algo, case, lo, agm)        # north_star's bar, in units of the frame maximum, at every level
                if hi <= 10:                                           # the strongest bins: the bar as a per-bin relative error
                    assert rgm <= 1e-5, (geom, algo, case, lo, rgm)
            e_gpu = (np.abs(cx[well] - truth[well]) / scale).max()
            e_cpu = (np.abs(wcx[well] - truth[well]) / scale).max()
            full = well & (fmax >= 0.1 * inp)                          # frames that fill their windows (not dominated by cancellation)
            e_full = (np.abs(cx[full] - truth[full]) / fmax[full][:, None]).max() if full.any() else 0.0
            report(OUT, f"{geom} {NAME[algo]} {case}: {int(well.sum())} frames with max|z| >= 1 % of the input scale: max abs err / frame max: "
                        f"gpu {e_gpu:.2e}, oracle {e_cpu:.2e}; {int(full.sum())} frames with max|z| >= 10 %: gpu {e_full:.2e}")
            assert e_gpu <= 4.0 * e_cpu + 2e-7, (geom, algo, case, e_gpu, e_cpu)
            assert e_full <= 2e-6, (geom, algo, case, e_full)
        ill = (~well) & (xp > 0)
        if ill.any():
            # stream start (and any other cancellation-dominated frame): errors against the frame maximum and against the input scale
            report(OUT, f"# {geom} {NAME[algo]} {case}: {int(ill.sum())} cancellation-dominated frames (max|z| < 1 % of sqrt(sr) max|x|): frame, "
                        f"max|z| / (sqrt(sr) max|x|), GPU err / max|z|, oracle err / max|z|, GPU err / (sqrt(sr) max|x|), oracle err / (sqrt(sr) max|x|)")
            for f in np.nonzero(ill)[0]:
                eg = np.abs(cx[f] - truth[f]).max()
                ec = np.abs(wcx[f] - truth[f]).max()
                if f < 40 or f % 16 == 0:
                    report(OUT, f"{geom} {NAME[algo]} {case} frame {f:4d}  {fmax[f] / inp[f]:.2e}  {eg / max(fmax[f], 1e-30):.2e}  "
                                f"{ec / max(fmax[f], 1e-30):.2e}  {eg / inp[f]:.2e}  {ec / inp[f]:.2e}")
                assert eg / inp[f] <= 1e-7, (geom, algo, case, f, eg / inp[f])   # = 1e-5 of the 1 % floor, the bar of test_parity_gpu.py (these frames see little of the signal)


def _decisive_margin(frame, b, bpo, ap):
    """distance (dB) of bin b of `frame` from the threshold that decides whether it is a peak: its height margin and, when it
    is a strict local maximum, its prominence margin (the smaller one decides)."""
    hb = ap.bass_min_height if b <= ap.highest_bassnote else ap.peak_min_height
    pb = ap.bass_min_prominence if b <= ap.highest_bassnote else ap.peak_min_prominence
    m = abs(float(frame[b]) - hb)
    if 0 < b < frame.size - 1 and frame[b] > frame[b - 1] and frame[b] > frame[b + 1]:
        prom = peak_prominences(frame.astype(np.float64), [b])[0][0]
        m = min(m, abs(prom - pb))
    else:   # not a strict local maximum here: a tie or a slope decides, i.e. a neighbour within the tolerance
        m = min(m, min(abs(float(frame[b]) - float(frame[b - 1])) if b > 0 else np.inf,
                       abs(float(frame[b]) - float(frame[b + 1])) if b < frame.size - 1 else np.inf))
    return m


@pytest.mark.parametrize("algo", [P.ALGO_FFT, P.ALGO_BLOCKDFT])
def test_end_to_end_peak_sets_every_difference_accounted(algo):
    """PCM -> dB -> peaks on the GPU against PCM -> dB -> peaks on the CPU (north_star: identical peak-bin indices).
    The peak logic is exact on identical frames (tests/test_peaks_gpu.py); end to end a set can differ only where a dB
    value lies within the dB parity tolerance (1e-2) of the deciding threshold.  Every such frame is listed."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    v.set_algo(algo)
    ov = O.OracleVqt(op)
    ap = O.OracleAnalysisParams()
    hop, nf, n_lead = 256, 4000, 20000
    n = n_lead + hop * nf
    t = np.arange(n) / op.sr
    pcm = white_noise(n, 8, amp=0.25).astype(np.float64)
    for k in (12, 19, 31, 40, 47):
        pcm += 0.1 * np.sin(2 * np.pi * 55.0 * 2 ** (k / 12.0) * t)
    pcm = pcm.astype(np.float32)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    words = (v.n_bins + 31) // 32
    d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
    v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, n_lead=n_lead)
    torch.cuda.synchronize()
    assert v.last_algo() == algo
    mask = d_mask.cpu().numpy().view(np.uint32)
    gdb = d_db.cpu().numpy()
    wdb = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead)
    n_peaks = 0
    diffs = []
    for f in range(nf):
        gp = mask_to_indices(mask[f], v.n_bins)
        wp = O.find_peaks_split(wdb[f], 36)
        n_peaks += wp.size
        assert np.array_equal(gp, O.find_peaks_split(gdb[f], 36)), f   # the peak logic itself: exact
        if not np.array_equal(gp, wp):
            for b in sorted(set(gp.tolist()) ^ set(wp.tolist())):
                diffs.append((f, int(b), float(gdb[f, b]), float(wdb[f, b]), _decisive_margin(wdb[f], int(b), 36, ap)))
    frames = sorted(set(d[0] for d in diffs))
    report(OUT, f"# end-to-end peak sets [{NAME[algo]}], 48 kHz / 252 bins, {nf} frames of noise + 5 tones, {n_peaks} oracle peaks: "
                f"{len(frames)} frames differ ({len(diffs)} bins); max |dB_gpu - dB_oracle| {np.abs(gdb - wdb).max():.2e}")
    for (f, b, g, w, m) in diffs:
        report(OUT, f"peakdiff [{NAME[algo]}] frame {f} bin {b}: dB gpu {g:.5f} oracle {w:.5f}; margin to the deciding threshold {m:.2e} dB")
        assert m <= 2e-2, (f, b, m)   # the bin's own dB and the base of its prominence may each move by the 1e-2 dB parity tolerance
    assert len(frames) <= nf // 200
