"""Many streams in ONE call (pvq_vqt_calculate_batch_db_streams / pvq_vqt_analyze_batch_streams): the reference's only batch driver
analyses many independent files side by side, one Vqt per rayon worker (pitchvis_train/src/train.rs:146-163), and BASELINE
configs[3] is stereo = two streams.  On the block-DFT path all streams share each stage's launch (tile-list entries carry a segment
index); every value must equal, BIT FOR BIT, what the single-stream entry point computes for that stream alone — dB rows, peak
masks, counts, continuous peaks — on every test geometry, with ragged lengths, leads (shard halos), sub-batching, both GEMM
arithmetics, and on the fall-back paths (general hop: FFT path, one launch per stream)."""
import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from helpers import GEOMS, get_geom, white_noise

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _streams(n_streams, hop, frames, leads, seed):
    out = []
    for s in range(n_streams):
        x = white_noise(leads[s] + hop * frames[s], seed + s, amp=0.25)
        t = np.arange(x.size) / 48000.0
        x = (x + 0.1 * np.sin(2 * np.pi * (220.0 * (1 + s % 5)) * t)).astype(np.float32)
        out.append(torch.from_numpy(x).cuda())
    return out


def _alloc(n_streams, stride, nb, max_peaks):
    words = (nb + 31) // 32
    return dict(db=torch.full((n_streams, stride, nb), -1.0, device="cuda"),
                mask=torch.full((n_streams, stride, words), -1, dtype=torch.int32, device="cuda"),
                cnt=torch.full((n_streams, stride), -1, dtype=torch.int32, device="cuda"),
                ctr=torch.zeros((n_streams, stride, max_peaks), device="cuda"),
                sz=torch.zeros((n_streams, stride, max_peaks), device="cuda"))


def _single(v, pcm, hop, nf, lead, nb, max_peaks):
    words = (nb + 31) // 32
    o = dict(db=torch.empty((nf, nb), device="cuda"), mask=torch.zeros((nf, words), dtype=torch.int32, device="cuda"),
             cnt=torch.zeros(nf, dtype=torch.int32, device="cuda"), ctr=torch.zeros((nf, max_peaks), device="cuda"),
             sz=torch.zeros((nf, max_peaks), device="cuda"))
    v.vqt_analyze_batch_device(pcm, hop, nf, o["db"], o["mask"], o["cnt"], o["ctr"], o["sz"], max_peaks, n_lead=lead)
    torch.cuda.synchronize()
    return o


def _check_equal(v, pcms, hop, frames, leads, stride, max_peaks=48):
    nb = v.n_bins
    o = _alloc(len(pcms), stride, nb, max_peaks)
    v.batch_streams_device(pcms, hop, frames, o["db"], stride, n_leads=leads, d_peak_mask=o["mask"], d_peak_count=o["cnt"],
                           d_center=o["ctr"], d_size=o["sz"], max_peaks=max_peaks)
    torch.cuda.synchronize()
    v.input_status()
    # the single-stream calls on the SAME path (left to itself ALGO_AUTO sends a stream of fewer than 64 frames to the FFT path, whose
    # values agree with the block-DFT path's to the parity bars, not bit for bit)
    v.set_algo(v.last_algo())
    for s in range(len(pcms)):
        nf = frames[s]
        if nf:
            w = _single(v, pcms[s], hop, nf, leads[s], nb, max_peaks)
            for k in ("db", "mask", "cnt", "ctr", "sz"):
                assert torch.equal(o[k][s, :nf], w[k]), (s, k, int((o[k][s, :nf] != w[k]).sum()))
        # rows a stream does not fill: zero frames, no peaks
        assert float(o["db"][s, nf:].abs().sum()) == 0.0 and int(o["cnt"][s, nf:].abs().sum()) == 0 and int(o["mask"][s, nf:].abs().sum()) == 0, s
    v.set_algo(P.ALGO_AUTO)
    return o


@pytest.mark.parametrize("name,hop", [("bench_48k_252", 256), ("bench_48k_288", 256), ("default_22k_588", 256), ("hires_96k_360", 128),
                                      ("hires_96k_840", 128), ("serial_22k_180", 256), ("bench_48k_252", 64), ("bench_48k_252", 1024),
                                      ("bench_48k_252", 1600), ("default_22k_588", 1344), ("bench_48k_252", 800)])   # (general hops, blockdft_gemm_gen; 800: two interleaved grids of 1 600)
def test_streams_equal_single_stream_calls_bit_for_bit(name, hop):
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    frames = [700, 64, 1, 300, 0, 257, 130, 999]          # ragged, one empty stream, one of a single frame
    leads = [0, 5000, 0, v.window_union - hop, 0, 123, 40000, 0]   # stream starts, shard halos, odd leads
    pcms = _streams(len(frames), hop, frames, leads, 1000)
    v.set_algo(P.ALGO_BLOCKDFT)   # (left to itself ALGO_AUTO sends 2 451 frames at a general hop to the path pvq_vqt_resolve_algo names: below)
    _check_equal(v, pcms, hop, frames, leads, stride=1024)
    assert v.last_algo() == P.ALGO_BLOCKDFT
    want = v.resolve_algo(hop, sum(frames))
    assert want == P.ALGO_BLOCKDFT or (hop & (hop - 1)) != 0
    _check_equal(v, pcms, hop, frames, leads, stride=1024)
    assert v.last_algo() == want


def test_streams_uniform_lengths_and_both_arithmetics():
    """64 streams x 512 frames, stride == n_frames (no padding rows, no memset), fp32 and split-bf16 GEMM on one handle"""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, n, nf = 256, 64, 512
    frames, leads = [nf] * n, [0 if s % 3 else 16128 for s in range(n)]
    pcms = _streams(n, hop, frames, leads, 2000)
    a = _check_equal(v, pcms, hop, frames, leads, stride=nf)
    v.set_gemm_precision(P.GEMM_BF16X3)
    _check_equal(v, pcms, hop, frames, leads, stride=nf)
    v.set_gemm_precision(P.GEMM_F32)
    b = _check_equal(v, pcms, hop, frames, leads, stride=nf)
    assert all(torch.equal(a[k], b[k]) for k in a)     # and back: same bits


def test_streams_sub_batching_and_long_streams():
    """a workspace limit that forces several launches: long streams are cut into sub-batches, short ones packed together"""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    v.set_workspace_limit(24 << 20)     # ~4 000 frames per launch at 5.6 KB per frame
    hop = 256
    frames = [9000, 100, 100, 4100, 3900, 64, 5000]
    leads = [0] * len(frames)
    pcms = _streams(len(frames), hop, frames, leads, 3000)
    _check_equal(v, pcms, hop, frames, leads, stride=9000)
    v.set_workspace_limit(0)            # back to the default: one launch
    _check_equal(v, pcms, hop, frames, leads, stride=9000)


def test_streams_odd_hop_on_the_fft_path_in_one_launch():
    """hop 735 (pitchvis_serial's 1 / 30 s at its own 22 050 Hz: odd, no multiple of it suits the block-DFT path): the FFT path takes all
    streams in ONE launch too (frames numbered through the streams, a binary search per frame finds its stream); same bits as
    stream-by-stream calls; 40 streams so that workgroups hold frames of different streams side by side"""
    pp, _ = get_geom("serial_22k_180")
    v = P.Vqt.new(pp, 0)
    hop = 735
    rng = np.random.default_rng(3)
    frames = [int(x) for x in rng.integers(0, 140, 40)]
    frames[5] = 0
    leads = [int(x) for x in rng.integers(0, 9000, 40)]
    pcms = _streams(len(frames), hop, frames, leads, 4000)
    _check_equal(v, pcms, hop, frames, leads, stride=160)
    assert v.last_algo() == P.ALGO_FFT


def test_streams_feed_the_analysis_batch_on_the_device():
    """PCM of N streams -> ONE pvq_vqt_calculate_batch_db_streams call -> pvq_analysis_batch_preprocess_device straight from its
    output tensor ([stream][frame][bin]): the reference's default pipeline for many streams with no host hop; checked against the
    oracle chain in tests/test_analysis_batch_gpu.py::test_pipeline_pcm_to_analysis_on_device, here: same bits as per-stream
    transform calls feeding the same batch object"""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, n, nf = 1024, 12, 280
    frames, leads = [nf] * n, [0] * n
    pcms = _streams(n, hop, frames, leads, 5000)
    d_db = torch.empty((n, nf, v.n_bins), device="cuda")
    v.batch_streams_device(pcms, hop, frames, d_db)
    d_ref = torch.empty_like(d_db)
    v.set_algo(v.last_algo())    # (a single stream of 280 frames is below the 384 from which PVQ_ALGO_AUTO takes the block-DFT path)
    for s in range(n):
        v.calculate_batch_db_device(pcms[s], hop, nf, d_ref[s])
    v.set_algo(P.ALGO_AUTO)
    torch.cuda.synchronize()
    assert torch.equal(d_db, d_ref)
    outs = []
    for src in (d_db, d_ref):
        b = P.AnalysisBatch(pp.range, n)
        o = {"x_vqt_smoothed": torch.zeros((n, nf, v.n_bins), device="cuda"), "peak_count": torch.zeros((n, nf), dtype=torch.int32, device="cuda"),
             "scene_calmness": torch.zeros((n, nf), device="cuda")}
        b.preprocess_device(src, nf, hop / 48000.0, o)
        torch.cuda.synchronize()
        outs.append(o)
    assert all(torch.equal(outs[0][k], outs[1][k]) for k in outs[0]) and int(outs[0]["peak_count"].sum()) > 0


def test_pcm_to_analysis_in_one_call():
    """pvq_analysis_batch_preprocess_pcm: the consumers' loop (update_vqt -> AnalysisState::preprocess per analysis) for many streams in
    one call == the two calls it is made of, and the batch object's own frame buffer == a caller's"""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, n, nf = 800, 10, 150     # the viewer's 60 analyses per second at 48 kHz
    pcms = _streams(n, hop, [nf] * n, [0] * n, 6000)
    keys = dict(x_vqt_smoothed=(n, nf, v.n_bins), calmness=(n, nf, v.n_bins), scene_calmness=(n, nf))
    def outs():
        o = {k: torch.zeros(shape, device="cuda") for k, shape in keys.items()}
        o["peak_count"] = torch.zeros((n, nf), dtype=torch.int32, device="cuda")
        o["center"] = torch.zeros((n, nf, 32), device="cuda"); o["size"] = torch.zeros((n, nf, 32), device="cuda")
        return o
    a, b, c = outs(), outs(), outs()
    d_db = torch.empty((n, nf, v.n_bins), device="cuda")
    v.batch_streams_device(pcms, hop, [nf] * n, d_db, nf)
    P.AnalysisBatch(pp.range, n).preprocess_device(d_db, nf, hop / 48000.0, a, max_peaks=32)
    P.AnalysisBatch(pp.range, n).preprocess_pcm(v, pcms, nf, hop, outputs=b, max_peaks=32)
    kept = torch.empty_like(d_db)
    P.AnalysisBatch(pp.range, n).preprocess_pcm(v, pcms, nf, hop, outputs=c, max_peaks=32, d_db=kept)
    torch.cuda.synchronize()
    assert torch.equal(kept, d_db) and int(a["peak_count"].sum()) > 0
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    with pytest.raises(P.PvqError):   # another range
        P.AnalysisBatch(P.VqtRange(55.0, 6, 36), n).preprocess_pcm(v, pcms, nf, hop)


def test_streams_argument_checks():
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    pcms = _streams(2, 256, [64, 64], [0, 0], 1)
    d_db = torch.empty((2, 64, v.n_bins), device="cuda")
    with pytest.raises(P.PvqError):   # a stream longer than the output stride
        v.batch_streams_device(pcms, 256, [64, 65], d_db, 64)
    with pytest.raises(P.PvqError):   # null stream pointer with frames
        v.batch_streams_device([pcms[0], 0], 256, [64, 64], d_db, 64)
    with pytest.raises(P.PvqError):   # no CPU fallback
        P.Vqt.new(pp, None).batch_streams_device(pcms, 256, [64, 64], d_db, 64)
    v.batch_streams_device([], 256, [], d_db, 64)    # nothing to do
    # a NaN in ONE stream raises the handle's flag, the other streams' rows are what they are alone
    bad = pcms[1].clone(); bad[3000] = float("nan")
    v.batch_streams_device([pcms[0], bad], 256, [64, 64], d_db, 64)
    with pytest.raises(P.PvqError) as e:
        v.input_status()
    assert e.value.status == 9
    ref = torch.empty((64, v.n_bins), device="cuda")
    v.calculate_batch_db_device(pcms[0], 256, 64, ref)
    torch.cuda.synchronize()
    assert torch.equal(d_db[0], ref)


@pytest.mark.parametrize("seed", range(18))
def test_streams_fuzz_bit_identical_to_single_calls(seed):
    """seeded random batches — geometry, hop (power-of-two, general, interleaved-grid and FFT-path hops), stream count, ragged
    lengths from 0 to a few thousand frames (short streams are staged, long ones run as segments of their own), leads, output
    stride, workspace limit — every one bit for bit what stream-by-stream calls give"""
    rng = np.random.default_rng(9000 + seed)
    name, hops = [("bench_48k_252", (256, 512, 128, 1600, 800, 320, 735)), ("serial_22k_180", (256, 64, 1024, 704, 735)),
                  ("hires_96k_360", (128, 256, 3200, 1000)), ("default_22k_588", (256, 1344, 2048, 367)),
                  ("hires_96k_840", (128, 1600, 441)), ("bench_48k_288", (256, 1600, 800, 100))][seed % 4 if seed < 12 else 4 + seed % 2]   # (840 bins: 84 per octave, the peak search's distance rule)
    hop = int(hops[int(rng.integers(0, len(hops)))])
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    n = int(rng.integers(1, 28))
    big = 1 + int(rng.integers(0, 3))   # a few streams beyond the staging threshold
    frames = []
    for s in range(n):
        kind = rng.integers(0, 10)
        if kind == 0: frames.append(0)
        elif kind == 1: frames.append(int(rng.integers(1, 5)))
        elif kind == 2 and big > 0 and hop <= 512:
            frames.append(int(rng.integers(2049, 3500)))
            big -= 1
        else: frames.append(int(rng.integers(5, 700)))
    leads = [int(rng.integers(0, 2 * v.window_union)) if rng.integers(0, 2) else 0 for _ in range(n)]
    stride = max(frames + [1]) + int(rng.integers(0, 3)) * 17
    if rng.integers(0, 3) == 0:
        v.set_workspace_limit(int(rng.integers(8, 64)) << 20)
    pcms = _streams(n, hop, frames, leads, 7000 + 50 * seed)
    _check_equal(v, pcms, hop, frames, leads, stride=stride, max_peaks=int(rng.integers(1, 70)))
