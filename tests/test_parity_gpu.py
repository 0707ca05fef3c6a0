"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Tolerances (fp32 path; north_star: magnitudes within 1e-5 relative of the CPU path):
  * complex coefficients: |z_gpu - z_cpu| <= 1e-5 * S for every bin, where S is the largest
    magnitude in that frame — or, for frames whose output is tiny compared with their input (e.g.
    the first frames after silence, where the signal sits only in the Hann tails and the
    coefficients are the residue of large cancelling terms), 1 % of the response a sine of the
    window's peak amplitude would give (0.01 * sqrt(sr) * max|x|): the error of ANY fp32
    evaluation scales with the input, not with the cancelled output;
  * magnitudes, per bin, relative: <= 1e-5 for every bin within 20 dB (1e-1) of the frame maximum
    (weaker bins carry the absolute error of two different fp32 FFT orderings; they are covered by
    the first bound and by the f64 check below);
  * against the exact-in-f64 transform of the same kernel the GPU error must stay within
    4x the CPU oracle's own fp32 error + 2e-7 of the frame maximum (the block-DFT path sums 256
    products per hop block where the FFT sums log2 N butterfly levels; both sit near 5e-7);
  * dB before the frame-relative clamp/shift (10 log10 |z|^2): <= 2e-4 dB for bins within 20 dB
    of the frame maximum; final dB: <= 2e-3 dB for those bins (in the shift branch, vqt.rs:946-947,
    every bin inherits the error of the frame's weakest bin) and <= 1e-2 dB everywhere
    (a bin 55 dB down has 1/560 of the maximum's magnitude, so a 3e-7 absolute error is 1.5e-3 dB).
"""
import os

import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from oracle import model_f64 as MF
from helpers import GEOMS, get_geom, three_regime, white_noise, sine_sweep, mask_to_indices

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALGOS = [P.ALGO_FFT, P.ALGO_BLOCKDFT, "blockdft_bf16x3"]


def _set_algo(v, algo):
    """algo is a pvq_algo value, or 'blockdft_bf16x3' = block-DFT path with the split-bf16 GEMM"""
    if algo == "blockdft_bf16x3":
        v.set_algo(P.ALGO_BLOCKDFT)
        v.set_gemm_precision(P.GEMM_BF16X3)
        return P.ALGO_BLOCKDFT
    v.set_algo(algo)
    v.set_gemm_precision(P.GEMM_F32)
    return algo


def _applicable(v, algo, hop, nf):
    if algo == P.ALGO_FFT:
        return True
    try:
        _set_algo(v, algo)
        d = torch.zeros(hop * 64 + 40000, device="cuda")
        o = torch.empty((64, v.n_bins), device="cuda")
        v.calculate_batch_db_device(d, hop, 64, o, n_lead=40000)
        torch.cuda.synchronize()
        return True
    except P.PvqError as e:
        if e.status == 7:
            return False
        raise


def run_gpu(v, pcm, hop, nf, n_lead=0, want_cplx=True):
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm, np.float32)).cuda()
    d_db = torch.full((nf, v.n_bins), -1.0, device="cuda")
    d_cx = torch.zeros((nf, v.n_bins, 2), device="cuda") if want_cplx else None
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead, d_out_cplx=d_cx)
    torch.cuda.synchronize()
    db = d_db.cpu().numpy()
    cx = d_cx.cpu().numpy().view(np.complex64)[..., 0] if want_cplx else None
    return db, cx


def input_peak(pcm, hop, nf, n_lead, union):
    """max |x| over the window union of each frame"""
    a = np.abs(np.asarray(pcm, np.float32))
    out = np.zeros(nf, np.float32)
    for f in range(nf):
        end = n_lead + (f + 1) * hop
        out[f] = a[max(end - union, 0):end].max()
    return out


def assert_parity(db, cx, wdb, wcx, truth=None, xpeak=None, sr=None):
    fmax = np.abs(wcx).max(axis=1, keepdims=True)
    fmax = np.maximum(fmax, 1e-30)
    well = np.ones(fmax.shape[0], bool)
    if xpeak is not None:
        floor = 0.01 * np.sqrt(sr) * np.asarray(xpeak, np.float32)[:, None]
        well = (fmax >= floor)[:, 0]
        fmax = np.maximum(fmax, floor)
    assert (np.abs(cx - wcx) / fmax).max() <= 1e-5
    if not well.all():  # cancellation-dominated frames: only the scaled bound above and a loose dB bound
        assert np.abs(db - wdb)[~well].max() <= 0.05
        db, cx, wdb, wcx, fmax = db[well], cx[well], wdb[well], wcx[well], fmax[well]
        truth = truth[well] if truth is not None else None
    strong = np.abs(wcx) >= 1e-1 * fmax
    rel = np.abs(np.abs(cx) - np.abs(wcx))[strong] / np.abs(wcx)[strong]
    assert rel.size == 0 or rel.max() <= 1e-5
    if truth is not None:
        e_gpu = (np.abs(cx - truth) / fmax).max()
        e_cpu = (np.abs(wcx - truth) / fmax).max()
        assert e_gpu <= 4.0 * e_cpu + 2e-7, (e_gpu, e_cpu)
    top = np.abs(wcx) >= 0.1 * fmax
    raw_g = 20 * np.log10(np.maximum(np.abs(cx[top].astype(np.complex128)), 1e-30))
    raw_w = 20 * np.log10(np.maximum(np.abs(wcx[top].astype(np.complex128)), 1e-30))
    assert np.abs(raw_g - raw_w).max(initial=0) <= 2e-4
    assert np.abs(db - wdb)[top].max(initial=0) <= 2e-3
    assert np.abs(db - wdb).max(initial=0) <= 1e-2


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("name", list(GEOMS))
def test_batch_parity_noise(name, algo):
    pp, op = get_geom(name)
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    hop = 128 if op.sr > 90000 else 256
    nf, n_lead = 72, 33000
    if not _applicable(v, algo, hop, nf):
        pytest.skip("block-DFT path not applicable to this geometry/hop")
    algo_id = _set_algo(v, algo)
    pcm = white_noise(n_lead + hop * nf, 0x5EED0001)
    db, cx = run_gpu(v, pcm, hop, nf, n_lead)
    assert v.last_algo() == algo_id
    wdb, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
    truth = MF.from_oracle_params(op, values_from=ov).batch_complex(pcm, hop, 8, n_lead=n_lead)
    assert_parity(db[:8], cx[:8], wdb[:8], wcx[:8], truth)
    assert_parity(db, cx, wdb, wcx)


@pytest.mark.parametrize("hop,nf", [(64, 1200), (128, 900), (512, 700), (1024, 600)])
def test_blockdft_other_hops_vs_oracle(hop, nf):
    """The block-DFT path at the hops the other tests do not use (they run 256, and 128 at 96 kHz): 64 = one double k group per tile
    and windows of 128 hop blocks (partial sums + blockdft_tree_finish), 512 / 1024 = the slice of E staged in two / four passes,
    1024 = the shortest window (a group without any tree level).  Batches long enough for tiles that lie wholly inside the stream
    (the 64-column wide tiles and the 16-byte operand loads), with a lead that is no multiple of anything."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    n_lead = 4321
    if not _applicable(v, P.ALGO_BLOCKDFT, hop, nf):
        pytest.skip("block-DFT path not applicable to this hop")
    algo_id = _set_algo(v, P.ALGO_BLOCKDFT)
    pcm = white_noise(n_lead + hop * nf, 0xB10C + hop)
    db, cx = run_gpu(v, pcm, hop, nf, n_lead)
    assert v.last_algo() == algo_id
    wdb, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
    assert_parity(db, cx, wdb, wcx)


@pytest.mark.parametrize("name,hop,nf", [("bench_48k_252", 1600, 700), ("bench_48k_252", 800, 900), ("bench_48k_252", 320, 1300), ("bench_48k_252", 1280, 520),
                                         ("bench_48k_252", 3200, 300), ("bench_48k_252", 400, 1100), ("bench_48k_288", 1600, 300), ("default_22k_588", 1600, 400),
                                         ("default_22k_588", 2240, 300), ("default_22k_588", 32, 2100), ("hires_96k_360", 3200, 300), ("serial_22k_180", 704, 400)])
def test_blockdft_general_hops_vs_oracle(name, hop, nf):
    """Hops the power-of-two block-DFT form cannot take, on the block-DFT path all the same:
      * a multiple of 64 that does not divide the windows — 1 600 samples = 30 analyses per second at 48 kHz, the cadence of
        pitchvis_serial/src/main.rs:41; 3 200, 1 280, 2 240, 704 — runs blockdft_gemm_gen (whole hop blocks + the window's remainder,
        Horner combine; windows shorter than the hop are the remainder GEMM alone);
      * a hop whose r-fold (r = 2, 4, ...) is such a multiple, or a power of two — 800 = the viewer's 60 analyses per second at 48 kHz
        (pitchvis_viewer/src/app/desktop_app.rs:18) -> 2 x 1 600, 320 -> 4 x 1 280, 400 -> 4 x 1 600, 32 -> 2 x 64 — runs r interleaved
        block grids of hop r * hop (grid i holds the frames i, i + r, ...),
    against the oracle's per-frame FFT route on the same PCM, with the same parity bars as every other path, from a stream start
    (zeros before it: range-checked tiles) and with a lead that is no multiple of anything."""
    pp, op = get_geom(name)
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    for n_lead in (0, 4321):
        algo_id = _set_algo(v, P.ALGO_BLOCKDFT)
        pcm = white_noise(n_lead + hop * nf, 0xB10C + hop)
        db, cx = run_gpu(v, pcm, hop, nf, n_lead)
        assert v.last_algo() == algo_id
        wdb, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
        assert_parity(db, cx, wdb, wcx, xpeak=input_peak(pcm, hop, nf, n_lead, v.window_union), sr=op.sr)
    # ALGO_AUTO takes this path for such a hop once the batch pays its launch floor back (a general hop's K loops are hop / 2 deep: a
    # few hundred frames are faster on the FFT path), and says beforehand which: the same bits as the forced call either way
    _set_algo(v, P.ALGO_AUTO)
    # (at the reference's default geometry the FFT path's per-window kernels run level with the general-hop block path — 0.035 against 0.038 us per
    # frame at hop 1 600, profiles/r05_auto_rule.txt — and AUTO stays on the FFT path at every size; elsewhere a large batch goes to the block path)
    assert v.resolve_algo(hop, 64) in (P.ALGO_FFT, P.ALGO_BLOCKDFT)
    assert v.resolve_algo(hop, 1 << 20) == P.ALGO_BLOCKDFT or (name == "default_22k_588" and hop >= 1344)
    want_algo = v.resolve_algo(hop, nf)
    db2, cx2 = run_gpu(v, pcm, hop, nf, n_lead)
    assert v.last_algo() == want_algo
    if want_algo == P.ALGO_BLOCKDFT:
        assert np.array_equal(db2, db)
    else:
        _set_algo(v, P.ALGO_FFT)
        db3, _ = run_gpu(v, pcm, hop, nf, n_lead)
        assert np.array_equal(db2, db3)
    # a hop with no such multiple (735 = pitchvis_serial's 1 / 30 s at 22 050 Hz: odd) stays on the FFT path, and says so when forced
    _set_algo(v, P.ALGO_BLOCKDFT)
    with pytest.raises(P.PvqError) as e:
        run_gpu(v, pcm[:735 * 64], 735, 64, 0)
    assert e.value.status == 7


@pytest.mark.parametrize("algo", ALGOS)
def test_three_regimes_and_stream_start(algo):
    """silent / clip-branch / shift-branch frames (vqt.rs:939-951), starting from an empty ring
    buffer (zeros before the stream: the first frames see partly-filled windows)."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    hop, nf = 256, 384
    if not _applicable(v, algo, hop, nf):
        pytest.skip("not applicable")
    _set_algo(v, algo)
    pcm = three_regime(hop * nf, op.sr, 3)
    db, cx = run_gpu(v, pcm, hop, nf, 0)
    wdb, wcx = ov.calculate_batch(pcm, hop, nf, want_complex=True)
    silent = (np.abs(wcx) == 0).all(axis=1)
    assert silent.sum() > 20
    assert (db[silent] == 0).all() and (np.abs(cx[silent]) == 0).all()   # exact
    xp = input_peak(pcm, hop, nf, 0, v.window_union)
    assert_parity(db[~silent], cx[~silent], wdb[~silent], wcx[~silent], xpeak=xp[~silent], sr=op.sr)
    raw_min = (10 * np.log10(np.maximum(np.abs(wcx) ** 2, 1e-12)) - 10 * np.log10(0.09)).min(axis=1)
    assert (raw_min[~silent] > 0).sum() > 20, "shift branch not exercised"


@pytest.mark.parametrize("algo", ALGOS)
def test_sweep_parity(algo):
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    hop, nf = 256, 375  # BASELINE config 1: 2.000 s at 48 kHz
    if not _applicable(v, algo, hop, nf):
        pytest.skip("not applicable")
    _set_algo(v, algo)
    pcm = sine_sweep(hop * nf, op.sr)
    db, cx = run_gpu(v, pcm, hop, nf, 0)
    wdb, wcx = ov.calculate_batch(pcm, hop, nf, want_complex=True)
    xp = input_peak(pcm, hop, nf, 0, v.window_union)
    assert_parity(db, cx, wdb, wcx, xpeak=xp, sr=op.sr)
    assert (db.argmax(axis=1)[100:] == wdb.argmax(axis=1)[100:]).all()


def test_ragged_and_edge_cases():
    pp, op = get_geom("serial_22k_180")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    # hop that is not a power of two, single frame, tiny lead, hop > window union, empty batch
    for hop, nf, n_lead in ((441, 5, 0), (441, 1, 17), (1, 3, 5000), (40000, 2, 123), (256, 0, 0)):
        pcm = white_noise(n_lead + hop * nf + 1, 21)
        got = v.calculate_batch_db(pcm, hop, nf, n_lead=n_lead)
        assert got.shape == (nf, v.n_bins)
        if nf:
            want = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead)
            assert np.abs(got - want).max() <= 1e-2
            assert v.last_algo() == P.ALGO_FFT or hop in (1, 256)


@pytest.mark.parametrize("algo", [P.ALGO_BLOCKDFT, "blockdft_bf16x3"])
def test_blockdft_tile_boundaries(algo):
    """Frame counts around the fused tiles' edges (a 256-row tile of the 16384-sample group holds 193 complete
    frames, a 64-frame kernel-product tile, ...), leads that are not multiples of 4 samples (dword- vs 16-byte-load
    tiles) and a second call on the same handle: block-DFT path == FFT path == oracle."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    hop = 256
    for nf, n_lead in ((1, 0), (2, 16129), (63, 3), (64, 40001), (65, 0), (192, 2), (193, 16128), (194, 7), (257, 33333),
                       (450, 1)):
        pcm = white_noise(n_lead + hop * nf, 1000 + nf)
        _set_algo(v, algo)
        db, cx = run_gpu(v, pcm, hop, nf, n_lead)
        assert v.last_algo() == P.ALGO_BLOCKDFT
        v.set_algo(P.ALGO_FFT)
        db_f, cx_f = run_gpu(v, pcm, hop, nf, n_lead)
        wdb, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
        xp = input_peak(pcm, hop, nf, n_lead, v.window_union)
        assert_parity(db, cx, wdb, wcx, xpeak=xp, sr=op.sr)
        fmax = np.maximum(np.abs(wcx).max(axis=1, keepdims=True), 0.01 * np.sqrt(op.sr) * xp[:, None])
        assert (np.abs(cx - cx_f) / fmax).max() <= 1e-5


def test_instant_api_matches_reference_semantics():
    for name in ("default_22k_588", "bench_48k_252"):
        pp, op = get_geom(name)
        v = P.Vqt.new(pp, 0)
        ov = O.OracleVqt(op)
        x = O.test_create_sines(op, [440.0, 554.37, 82.41])
        a = v.calculate_vqt_instant_in_db(x)
        b = ov.calculate_vqt_instant_in_db(x)
        assert a.shape == (pp.range.n_buckets(),)
        assert np.abs(a - b).max() <= 1e-2 and np.abs(a - b)[b > b.max() - 20].max() <= 2e-4
        with pytest.raises(AssertionError):  # vqt.rs:867-871
            v.calculate_vqt_instant_in_db(x[:-1])
        from pitchvis_amd import _lib
        import ctypes as C
        out = np.zeros(v.n_bins, np.float32)
        st = _lib.load().pvq_vqt_calculate_instant_db(v._h, x.ctypes.data_as(C.POINTER(C.c_float)), 100,
                                                      out.ctypes.data_as(C.POINTER(C.c_float)))
        assert st == _lib.PVQ_ERR_BAD_LENGTH


@pytest.mark.parametrize("algo", ALGOS)
def test_golden_fixtures(algo):
    z = np.load(os.path.join(G, "bench_48k_252_frames.npz"))
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, nf, n_lead = int(z["hop"]), int(z["n_frames"]), int(z["n_lead"])
    if not _applicable(v, algo, hop, 64):
        pytest.skip("not applicable")
    _set_algo(v, algo)
    for case in ("noise", "regimes"):
        db, cx = run_gpu(v, z[f"{case}_pcm"], hop, nf, n_lead)
        keep = np.abs(z[f"{case}_cplx"]).max(axis=1) > 0
        xp = input_peak(z[f"{case}_pcm"], hop, nf, n_lead, v.window_union)
        assert_parity(db[keep], cx[keep], z[f"{case}_db"][keep], z[f"{case}_cplx"][keep], xpeak=xp[keep], sr=op.sr)
    z2 = np.load(os.path.join(G, "default_22k_588_frames.npz"))
    pp, op = get_geom("default_22k_588")
    v = P.Vqt.new(pp, 0)
    got = v.calculate_batch_db(z2["pcm"], int(z2["hop"]), int(z2["n_frames"]))
    assert np.abs(got - z2["db"]).max() <= 1e-2


@pytest.mark.parametrize("algo", ALGOS)
def test_full_size_properties(algo):
    """BASELINE config 2 size (65 536 hops of white noise, 48 kHz / 252 bins) through
    size-independent properties: homogeneity (x2 input => exactly x2 coefficients: scaling by a
    power of two is exact in fp32 through every linear stage), shift consistency (dropping k hops
    from the front leaves the later frames bit-identical), determinism, and a spot check of 48
    scattered frames against the oracle."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, nf = 256, 65536
    if not _applicable(v, algo, hop, nf):
        pytest.skip("not applicable")
    _set_algo(v, algo)
    g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0001)
    d_pcm = (torch.rand(hop * nf, device="cuda", generator=g) - 0.5) * 0.5
    d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, d_out_cplx=d_cx); torch.cuda.synchronize()
    # determinism
    d_db2 = torch.empty_like(d_db); d_cx2 = torch.empty_like(d_cx)
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db2, d_out_cplx=d_cx2); torch.cuda.synchronize()
    assert torch.equal(d_db, d_db2) and torch.equal(d_cx, d_cx2)
    # homogeneity
    v.calculate_batch_db_device(d_pcm * 2.0, hop, nf, d_db2, d_out_cplx=d_cx2); torch.cuda.synchronize()
    assert torch.equal(d_cx2, d_cx * 2.0)
    # shift consistency: stream without its first k hops, but carrying them as history
    k = 4096
    nf2 = nf - k
    v.calculate_batch_db_device(d_pcm, hop, nf2, d_db2, n_lead=k * hop, d_out_cplx=d_cx2); torch.cuda.synchronize()
    assert torch.equal(d_cx2[:nf2], d_cx[k:]) and torch.equal(d_db2[:nf2], d_db[k:])
    # spot check against the oracle
    ov = O.OracleVqt(op)
    pcm = d_pcm.cpu().numpy()
    rng = np.random.default_rng(1)
    frames = np.concatenate([[0, 1, 63, 64, nf - 1], rng.integers(65, nf - 1, 43)])
    db = d_db.cpu().numpy(); cx = d_cx.cpu().numpy().view(np.complex64)[..., 0]
    for f in frames:
        end = (f + 1) * hop
        beg = max(end - op.n_fft, 0)
        x = np.zeros(op.n_fft, np.float32); x[op.n_fft - (end - beg):] = pcm[beg:end]
        wcx = ov.calculate_vqt_instant_complex(x)[None]; wdb = ov.calculate_vqt_instant_in_db(x)[None]
        assert_parity(db[f:f + 1], cx[f:f + 1], wdb, wcx, xpeak=np.array([np.abs(x).max()]), sr=op.sr)
    assert torch.isfinite(d_db).all() and (d_db >= 0).all() and (d_db <= 60.0).all()


@pytest.mark.parametrize("hop", [64, 128, 512, 1024])
def test_blockdft_other_hops(hop):
    """Block-DFT path at other power-of-two hops: 64 makes the longest window 256 blocks (two-level tree), 1024 makes
    the shortest window a single block (no tree level at all); block-DFT == FFT path == oracle."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    ov = O.OracleVqt(op)
    nf, n_lead = 300, 777
    pcm = white_noise(n_lead + hop * nf, 4242 + hop)
    for algo in (P.ALGO_BLOCKDFT, "blockdft_bf16x3"):
        _set_algo(v, algo)
        db, cx = run_gpu(v, pcm, hop, nf, n_lead)
        assert v.last_algo() == P.ALGO_BLOCKDFT
        wdb, wcx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
        xp = input_peak(pcm, hop, nf, n_lead, v.window_union)
        assert_parity(db, cx, wdb, wcx, xpeak=xp, sr=op.sr)


@pytest.mark.parametrize("hop", [64, 256])
def test_blockdft_more_than_one_sub_batch(hop):
    """More frames than one sub-batch (the workspace, the stream rebasing and — at hop 64 — the partial-sum buffer of the
    two-level tree are reused per sub-batch): frames around the seam and at the end against the FFT path and the oracle.
    Sub-batches hold what the handle's workspace limit allows (131 072 frames at the default 1 GiB here); the limit is set
    so that they hold 65 536 to keep the test small (test_configs_gpu.py runs a full-size shard at the default)."""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    v.set_algo(P.ALGO_BLOCKDFT)
    warm = torch.zeros(hop * 64, device="cuda"); wdb_ = torch.empty((64, v.n_bins), device="cuda")
    v.calculate_batch_db_device(warm, hop, 64, wdb_); torch.cuda.synchronize()   # builds the tables: the column count is known
    per_frame = (v.blockdft_columns() + 32) * 8 * (2 if hop == 64 else 1)
    v.set_workspace_limit(65536 * per_frame + per_frame)
    ov = O.OracleVqt(op)
    nf, n_lead = 65536 + 700, 5
    pcm = white_noise(n_lead + hop * nf, 99 + hop)
    d_pcm = torch.from_numpy(pcm).cuda()
    outs = {}
    for algo in (P.ALGO_BLOCKDFT, P.ALGO_FFT):
        _set_algo(v, algo)
        d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
        v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead, d_out_cplx=d_cx); torch.cuda.synchronize()
        assert v.last_algo() == algo
        outs[algo] = (d_db, d_cx)
    sel = np.concatenate([np.arange(65536 - 300, 65536 + 300), np.arange(nf - 40, nf), np.arange(0, 40)])
    a = outs[P.ALGO_BLOCKDFT][1][sel].cpu().numpy().view(np.complex64)[..., 0]
    b = outs[P.ALGO_FFT][1][sel].cpu().numpy().view(np.complex64)[..., 0]
    xp = input_peak(pcm, hop, nf, n_lead, v.window_union)[sel]
    fmax = np.maximum(np.abs(b).max(axis=1, keepdims=True), 0.01 * np.sqrt(op.sr) * xp[:, None])
    assert (np.abs(a - b) / fmax).max() <= 1e-5
    for f in (65535, 65536, 65537, nf - 1):
        end = n_lead + (f + 1) * hop
        beg = max(end - op.n_fft, 0)
        x = np.zeros(op.n_fft, np.float32); x[op.n_fft - (end - beg):] = pcm[beg:end]
        wcx = ov.calculate_vqt_instant_complex(x)[None]; wdb = ov.calculate_vqt_instant_in_db(x)[None]
        db = outs[P.ALGO_BLOCKDFT][0][f:f + 1].cpu().numpy(); cx = outs[P.ALGO_BLOCKDFT][1][f:f + 1].cpu().numpy().view(np.complex64)[..., 0]
        assert_parity(db, cx, wdb, wcx, xpeak=np.array([np.abs(x).max()]), sr=op.sr)
