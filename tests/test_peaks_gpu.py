"""GPU peak / note detection (K4) against the oracle's restatement of find_peaks 0.1.5 +
enhance_peaks_continuous + promote_bass_peaks_with_harmonics.

Bar: peak-bin indices BIT-IDENTICAL when both sides see the same dB frame; continuous centre
within max(1e-4 bins, 4 ulp) (device expf/log2f differ from glibc by <= 2 ulp; 1 ulp at bin 800 is 6e-5) and size
within 2e-3 dB (the size is interpolated at the centre: its error is the centre error times the local
slope of up to ~10 dB/bin).
End to end (GPU dB -> GPU peaks vs CPU dB -> CPU peaks) the sets are identical except in frames
where a dB value sits within the dB parity tolerance of a threshold or a tie; those frames are
counted and listed, not hidden."""
import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from helpers import get_geom, white_noise, mask_to_indices, report

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _frames(op, nf, seed, hop=2048):
    ov = O.OracleVqt(op)
    rng = np.random.default_rng(seed)
    n = hop * nf + 30000
    t = np.arange(n) / op.sr
    pcm = white_noise(n, seed, amp=0.05).astype(np.float64)
    for k in rng.integers(0, op.octaves * 12 - 6, 6):
        pcm += 0.08 * np.sin(2 * np.pi * op.min_freq * 2 ** (k / 12.0) * t * (1 + 0.002 * rng.standard_normal()))
    return ov.calculate_batch(pcm.astype(np.float32), hop, nf, n_lead=30000)


@pytest.mark.parametrize("name", ["bench_48k_252", "bench_48k_288", "default_22k_588", "hires_96k_360", "hires_96k_840", "serial_22k_180"])
def test_peaks_bit_identical_on_same_frames(name):
    pp, op = get_geom(name)
    v = P.Vqt.new(pp, 0)
    db = np.concatenate([_frames(op, 96, 5), np.abs(np.random.default_rng(3).normal(0, 9, (160, v.n_bins))).astype(np.float32)])
    mask, count, center, size = v.analyze_batch(db, max_peaks=160)
    for f in range(db.shape[0]):
        wp, wce, wsz = O.analyze_frame(db[f], op.min_freq, op.octaves, op.buckets_per_octave)
        gp = mask_to_indices(mask[f], v.n_bins)
        assert np.array_equal(gp, wp), (name, f)
        assert count[f] == wp.size
        k = wp.size
        assert (np.abs(center[f, :k] - wce) <= np.maximum(1e-4, 4 * np.spacing(np.abs(wce).astype(np.float32)))).all()
        ctol = np.maximum(1e-4, 4 * np.spacing(np.abs(wce).astype(np.float32)))
        assert (np.abs(size[f, :k] - wsz) <= 2e-3 + 40.0 * ctol).all()  # up to ~40 dB/bin slope in random frames


@pytest.mark.parametrize("name", ["bench_48k_252", "bench_48k_288", "default_22k_588", "hires_96k_840"])
def test_random_ties_two_and_three_sample_plateaus(name):
    """Exact ties of neighbouring bins (the lean kernel takes two-sample plateaus itself, longer ones go to the
    generic kernel): random frames with ties copied in at random places, at the frame edges and next to each other."""
    pp, op = get_geom(name)
    v = P.Vqt.new(pp, 0)
    n = v.n_bins
    rng = np.random.default_rng(17)
    frames = np.abs(rng.normal(0, 9, (240, n))).astype(np.float32)
    for f in range(frames.shape[0]):
        for _ in range(rng.integers(1, 12)):
            i = int(rng.integers(0, n - 3))
            ln = 2 if f % 3 else int(rng.integers(2, 5))
            frames[f, i:i + ln] = frames[f, i]
        if f % 7 == 0:
            frames[f, 0:2] = frames[f, 0]              # tie at the left edge
        if f % 11 == 0:
            frames[f, n - 2:n] = 35.0                  # tie at the right edge: never a peak
        if f % 13 == 0:
            frames[f, 40:42] = 30.0; frames[f, 42:44] = 31.0   # rising plateau then a plateau peak
    mask, count, center, size = v.analyze_batch(frames, max_peaks=n)
    for f in range(frames.shape[0]):
        wp, wce, wsz = O.analyze_frame(frames[f], op.min_freq, op.octaves, op.buckets_per_octave)
        assert np.array_equal(mask_to_indices(mask[f], n), wp), (name, f)
        assert count[f] == wp.size


@pytest.mark.parametrize("bpo,octaves", [(120, 5), (144, 4), (60, 7)])
def test_distance_rule_at_other_distances(bpo, octaves):
    """min_distance = round(0.4 bpo / 12): 4 at 120 bins per octave (the register form of the rule, pk_distance_regs: at most one other candidate within reach
    on each side), 5 at 144 (two within reach: the rounds over LDS, pk_distance_rounds), 2 at 60; peak sets against the oracle on random frames with and without ties"""
    from helpers import geom_pair
    pp, op = geom_pair(48000.0, 55.0, octaves, bpo)
    v = P.Vqt.new(pp, 0)
    n = v.n_bins
    rng = np.random.default_rng(bpo)
    frames = np.abs(rng.normal(0, 9, (200, n))).astype(np.float32)
    frames[100:] = np.round(frames[100:] * 2.0) / 2.0     # a coarse grid: many exact ties between neighbouring candidates (the tie rule: the later position wins)
    for f in range(0, 200, 5):
        i = int(rng.integers(4, n - 12))
        frames[f, i:i + 9:2] = 30.0 + np.arange(5, dtype=np.float32) * (0.0 if f % 10 else 0.5)   # a chain of candidates two bins apart
        frames[f, i + 1:i + 8:2] = 1.0
    mask, count, center, size = v.analyze_batch(frames, max_peaks=n)
    for f in range(frames.shape[0]):
        wp, wce, wsz = O.analyze_frame(frames[f], op.min_freq, op.octaves, op.buckets_per_octave)
        assert np.array_equal(mask_to_indices(mask[f], n), wp), (bpo, f)
        assert count[f] == wp.size


def test_crafted_plateaus_edges_and_split():
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    n = v.n_bins
    frames = np.zeros((6, n), np.float32)
    frames[0, [0, n - 1]] = 30.0                      # edges are never peaks
    frames[1, 50:53] = 20.0                           # plateau -> middle
    frames[1, 100:102] = 12.0                         # even plateau
    frames[2, [1, 2, 3]] = [9.0, 9.5, 9.0]            # min_bin = 2
    frames[2, 1] = 0.0; frames[2, 2] = 9.0; frames[2, 3] = 0.0
    frames[3, 28] = 6.0; frames[3, 30] = 6.0; frames[3, 60] = 10.0; frames[3, 90] = 3.99  # split at 28
    frames[4] = 5.0                                   # flat
    frames[5] = np.linspace(0, 40, n)                 # monotone
    fa = v.analyze_frames(frames)
    for f in range(frames.shape[0]):
        wp, _, _ = O.analyze_frame(frames[f], op.min_freq, op.octaves, op.buckets_per_octave)
        assert sorted(fa[f].peaks) == list(wp), f
    assert sorted(fa[1].peaks) == [51, 101] and sorted(fa[3].peaks) == [28, 60] and not fa[0].peaks


def test_vqt_close_frequencies_end_to_end_on_gpu():
    """reference lib.rs:16-48 through the GPU: two sines a semitone apart => exactly 2 peaks."""
    pp, op = get_geom("default_22k_588")
    v = P.Vqt.new(pp, 0)
    sub = 30
    stim = []
    for i in range(int(2.6 * sub), op.octaves * sub - sub // 2):
        ln = np.float32(i) / np.float32(sub)
        f1 = np.float32(op.min_freq) * np.float32(2.0) ** ln
        f2 = np.float32(op.min_freq) * np.float32(2.0) ** (ln + np.float32(1.0 / 12.0))
        stim.append(O.test_create_sines(op, [f1, f2]))
    pcm = np.concatenate(stim)
    db = v.calculate_batch_db(pcm, op.n_fft, len(stim))       # one n_fft buffer per frame
    horizon_ms = (70.0 * (1.5 - 0.5 * (np.arange(v.n_bins, dtype=np.float32) / op.buckets_per_octave / op.octaves)) * 0.6).astype(np.int64)
    alpha = (1.0 - np.exp(-2.0 * 1.1 / (horizon_ms / 1000.0))).astype(np.float32)
    fa = v.analyze_frames((db * alpha[None, :]).astype(np.float32))
    assert [len(f.peaks) for f in fa] == [2] * len(stim)


def test_end_to_end_peak_sets_with_near_threshold_accounting():
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    hop, nf, n_lead = 256, 1500, 20000
    rng = np.random.default_rng(8)
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
    mask = d_mask.cpu().numpy().view(np.uint32)
    gdb = d_db.cpu().numpy()
    wdb = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead)
    differing = []
    for f in range(nf):
        gp = mask_to_indices(mask[f], v.n_bins)
        # the peak logic itself must be exact on the GPU's own frame ...
        assert np.array_equal(gp, O.find_peaks_split(gdb[f], 36)), f
        # ... and end to end the sets agree unless a threshold is within the dB tolerance
        if not np.array_equal(gp, O.find_peaks_split(wdb[f], 36)):
            differing.append(f)
    report("parity_evidence_r02.txt", f"# tests/test_peaks_gpu.py end-to-end: peak sets differ in {len(differing)} of {nf} frames "
                                      f"(near-threshold; each accounted for in test_parity_evidence_gpu.py): {differing[:20]}")
    assert len(differing) <= nf // 100
