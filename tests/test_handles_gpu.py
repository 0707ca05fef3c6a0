"""SURVEY 8b, ownership and threading: `Vqt` takes `&mut self` (one caller at a time per instance), instances are independent and live one per
worker thread (pitchvis_train/src/train.rs:148-154: a `Vqt` per rayon worker), and Bevy calls the viewer's instance from whatever
scheduler thread runs the system (pitchvis_viewer/src/vqt_system.rs:5-6: a `Resource`, so `Send + Sync`).  The C ABI's promise is
the same: a handle is not thread-safe, handles are independent.  Here: worker threads with a handle each (different geometries,
hops and paths) hammer the library at the same time and must reproduce, bit for bit, what the same calls give one after the other;
a handle made on one thread works on another; error texts stay with the thread that caused them."""
import threading

import numpy as np
import pytest

import pitchvis_amd as P
from helpers import get_geom, white_noise

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

JOBS = [("bench_48k_252", 256, 3000, P.ALGO_AUTO), ("default_22k_588", 256, 1500, P.ALGO_AUTO), ("bench_48k_252", 1600, 2500, P.ALGO_BLOCKDFT),
        ("serial_22k_180", 735, 400, P.ALGO_AUTO), ("hires_96k_360", 128, 1200, P.ALGO_AUTO), ("bench_48k_288", 800, 700, P.ALGO_FFT)]


def _run(v, pcm, hop, nf, rounds):
    """`rounds` analyze calls on a stream of the caller's own (the buffers are made and read back on that stream too: torch's fills run on
    its current stream, and nothing orders another stream behind them); every round's outputs"""
    words = (v.n_bins + 31) // 32
    st = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(st):
        for _ in range(rounds):
            o = dict(db=torch.empty((nf, v.n_bins), device="cuda"), mask=torch.zeros((nf, words), dtype=torch.int32, device="cuda"),
                     cnt=torch.zeros(nf, dtype=torch.int32, device="cuda"), ctr=torch.zeros((nf, 48), device="cuda"), sz=torch.zeros((nf, 48), device="cuda"))
            v.vqt_analyze_batch_device(pcm, hop, nf, o["db"], o["mask"], o["cnt"], o["ctr"], o["sz"], 48, n_lead=123, stream=st)
            outs.append(o)
        st.synchronize()
        return [{k: t.cpu().numpy() for k, t in o.items()} for o in outs]


def test_a_handle_per_thread_all_at_once_equals_one_after_the_other():
    made = []
    for name, hop, nf, algo in JOBS:
        pp, _ = get_geom(name)
        v = P.Vqt.new(pp, 0)          # made on the main thread, used on a worker: handles are not tied to their creating thread
        v.set_algo(algo)
        pcm = torch.from_numpy(white_noise(123 + hop * nf, 77 + hop)).cuda()
        made.append((v, pcm, hop, nf))
    torch.cuda.synchronize()
    serial = [_run(v, pcm, hop, nf, 1)[0] for v, pcm, hop, nf in made]
    results, errors = [None] * len(made), []

    def work(i):
        try:
            v, pcm, hop, nf = made[i]
            results[i] = _run(v, pcm, hop, nf, 6)
        except Exception as e:   # noqa: BLE001  (handed to the main thread)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(made))]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for i, rounds in enumerate(results):
        for o in rounds:
            for k, a in serial[i].items():
                assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, o[k].view(np.uint32) if o[k].dtype == np.float32 else o[k]), (JOBS[i], k)


def test_error_text_stays_with_its_thread():
    """pvq_last_error is per thread: a failing call on one thread does not overwrite what another thread reads"""
    pp, _ = get_geom("serial_22k_180")
    v1, v2 = P.Vqt.new(pp, 0), P.Vqt.new(pp, 0)
    seen = {}

    def bad():
        v1.set_algo(P.ALGO_BLOCKDFT)
        pcm = torch.zeros(735 * 64, device="cuda")
        db = torch.empty((64, v1.n_bins), device="cuda")
        try:
            v1.calculate_batch_db_device(pcm, 735, 64, db)     # no multiple of 735 suits the block-DFT path
        except P.PvqError as e:
            seen["bad"] = (e.status, str(e))

    def good():
        pcm = torch.zeros(256 * 64, device="cuda")
        db = torch.empty((64, v2.n_bins), device="cuda")
        v2.calculate_batch_db_device(pcm, 256, 64, db)
        torch.cuda.synchronize()
        seen["good"] = P.last_error() if hasattr(P, "last_error") else ""

    t1 = threading.Thread(target=bad); t1.start(); t1.join()
    t2 = threading.Thread(target=good); t2.start(); t2.join()
    assert seen["bad"][0] == 7 and "block-DFT" in seen["bad"][1]
    assert "block-DFT" not in seen["good"]


def test_handles_swapped_wholesale_do_not_leak_device_memory():
    """the viewer replaces its `Vqt` at run time when the parameters change (pitchvis_viewer/src/app/common.rs:1130-1133: the old one
    is dropped): create -> use on every path (power-of-two hop, general hop, FFT path, a staged many-streams call, the device
    `AnalysisState` batch) -> destroy, thirty times over; the device's free memory afterwards is what it was after the first
    round (workspaces, tile lists, tables, shard buffers and streams all go with the handle)"""
    import gc
    pp, _ = get_geom("bench_48k_252")

    def one_round():
        v = P.Vqt.new(pp, 0)
        nf = 3000
        pcm = torch.from_numpy(white_noise(1600 * nf, 5)).cuda()
        db = torch.empty((nf, v.n_bins), device="cuda")
        for hop, algo in ((256, P.ALGO_AUTO), (1600, P.ALGO_BLOCKDFT), (735, P.ALGO_AUTO)):
            v.set_algo(algo)
            v.calculate_batch_db_device(pcm, hop, nf, db)
        v.set_algo(P.ALGO_AUTO)
        sdb = torch.empty((6, 500, v.n_bins), device="cuda")
        v.batch_streams_device([pcm[i * 200000: i * 200000 + 256 * 500] for i in range(6)], 256, [500] * 6, sdb)
        ab = P.AnalysisBatch(P.VqtRange(55.0, 7, 36), n_streams=6)
        ab.preprocess_device(sdb, 500, frame_time=256 / 48000.0)
        torch.cuda.synchronize()
        v.input_status()
        del ab, v, pcm, db, sdb
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    first = one_round()
    for _ in range(29):
        last = one_round()
    assert first - last < (8 << 20), f"free device memory fell by {(first - last) / 2**20:.1f} MiB over 29 create / use / destroy rounds"


@pytest.mark.parametrize("name,hop", [("bench_48k_252", 800), ("default_22k_588", 735), ("hires_96k_360", 1000), ("serial_22k_180", 441), ("bench_48k_288", 256)])
def test_few_frames_on_the_fft_path_equal_the_same_frames_of_a_batch(name, hop):
    """A call of a few frames (the streaming front end's single frame, a short clip) runs the FFT path group-split — a workgroup per
    window group, the frames finished by db_rows — where a batch walks the groups inside one workgroup: same bits, dB rows and
    complex coefficients, for 1 ... 40 frames cut out of a 300-frame batch at odd places"""
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    v.set_algo(P.ALGO_FFT)
    nf, lead = 300, 1234
    pcm = torch.from_numpy(white_noise(lead + hop * nf, 31 + hop)).cuda()
    db = torch.empty((nf, v.n_bins), device="cuda"); cx = torch.empty((nf, v.n_bins, 2), device="cuda")
    v.calculate_batch_db_device(pcm, hop, nf, db, n_lead=lead, d_out_cplx=cx)
    for first, n in ((0, 1), (0, 3), (137, 1), (50, 17), (200, 40), (299, 1)):
        sdb = torch.empty((n, v.n_bins), device="cuda"); scx = torch.empty((n, v.n_bins, 2), device="cuda")
        # the same frames as a call of their own: everything before their first hop is lead
        v.calculate_batch_db_device(pcm[: lead + hop * (first + n)], hop, n, sdb, n_lead=lead + hop * first, d_out_cplx=scx)
        torch.cuda.synchronize()
        assert torch.equal(sdb, db[first:first + n]) and torch.equal(scx, cx[first:first + n]), (first, n)
    v.input_status()


def test_resolve_algo_is_a_pure_query():
    """pvq_vqt_resolve_algo answers from the plan alone: before the first batch it builds no block-DFT tables (blockdft_columns stays 0 — it used to
    build those of hop * r on whatever device was current), and between two batches it leaves the tables of the hop in use where they are."""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    assert v.blockdft_columns() == 0
    assert v.resolve_algo(800, 100000) == P.ALGO_BLOCKDFT and v.resolve_algo(800, 64) == P.ALGO_FFT
    assert v.resolve_algo(320, 100000) == P.ALGO_BLOCKDFT and v.resolve_algo(735, 100000) == P.ALGO_FFT
    assert v.blockdft_columns() == 0
    nf = 2000
    pcm = torch.from_numpy(white_noise(256 * nf, 5)).cuda()
    a, b = torch.empty((nf, v.n_bins), device="cuda"), torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(pcm, 256, nf, a)
    cols = v.blockdft_columns()
    assert cols > 0
    assert v.resolve_algo(800, 100000) == P.ALGO_BLOCKDFT
    assert v.blockdft_columns() == cols
    v.calculate_batch_db_device(pcm, 256, nf, b)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_single_frame_call_behind_an_asynchronous_batch_on_another_stream():
    """One handle's calls are ordered (pvq.h): calculate_vqt_instant_in_db runs on a stream of the handle's own and shares the handle's
    workspaces with a batch that may still be queued on the caller's stream.  A few-frame batch on the group-split FFT route (it writes the same
    scratch rows the single-frame route does) is queued behind a long block-DFT batch on a user stream and, without waiting, the single-frame
    call follows; both must give what they give alone."""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    n_fft = pp.n_fft
    x = white_noise(n_fft, 901)
    alone = v.calculate_vqt_instant_in_db(x)
    nf_long, nf_few = 60000, 24
    pcm_long = torch.from_numpy(white_noise(256 * nf_long, 11)).cuda()
    pcm_few = torch.from_numpy(white_noise(256 * nf_few + 300, 12)).cuda()
    ref_few = torch.empty((nf_few, v.n_bins), device="cuda")
    v.calculate_batch_db_device(pcm_few, 256, nf_few, ref_few, n_lead=300)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    for _ in range(5):
        out_long = torch.empty((nf_long, v.n_bins), device="cuda")
        out_few = torch.full((nf_few, v.n_bins), -1.0, device="cuda")
        torch.cuda.synchronize()
        v.calculate_batch_db_device(pcm_long, 256, nf_long, out_long, stream=st)
        v.calculate_batch_db_device(pcm_few, 256, nf_few, out_few, n_lead=300, stream=st)
        got = v.calculate_vqt_instant_in_db(x)          # returns while `st` may still be busy
        assert np.array_equal(got.view(np.uint32), alone.view(np.uint32))
        st.synchronize()
        assert torch.equal(out_few.view(torch.int32), ref_few.view(torch.int32))


@pytest.mark.parametrize("hop", [256, 1600])
def test_host_batch_in_parts_takes_one_path(hop):
    """calculate_batch_db pipelines a large host batch in 16 384-frame parts; the path is resolved once for the whole call, so a tail part below
    PVQ_ALGO_AUTO's threshold does not switch to the FFT path: the host call equals the device call of the whole batch bit for bit."""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    nf = 2 * 16384 + 500
    pcm = white_noise(hop * nf + 77, 31 + hop)
    host = v.calculate_batch_db(pcm, hop, nf, n_lead=77)
    assert v.last_algo() == P.ALGO_BLOCKDFT
    d = torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(torch.from_numpy(pcm).cuda(), hop, nf, d, n_lead=77)
    torch.cuda.synchronize()
    assert np.array_equal(host.view(np.uint32), d.cpu().numpy().view(np.uint32))
