"""The multi-device driver behind the C ABI (pvq_plan_shard, pvq_vqt_analyze_batch_multi; SURVEY.md 8e, the rayon map_init pattern of
pitchvis_train/src/train.rs:146-155 for one long stream).  The shard planner is host arithmetic (no GPU); the driver is exercised on
the one-GPU box with k handles on device 0: k = 1 .. 8 virtual shards must reproduce the unsharded result bit for bit."""
import ctypes as C

import numpy as np
import pytest

import pitchvis_amd as P
from pitchvis_amd import _lib
from pitchvis_amd.sharding import plan_shard
from helpers import get_geom, white_noise


def _plan_py(n_total, hop, wu, rank, world):   # the arithmetic, restated
    base, extra = divmod(n_total, world)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    hop_begin = first * hop
    halo = max(wu - hop, 0)
    begin = max(hop_begin - halo, 0)
    return first, n, begin, hop_begin + n * hop, hop_begin - begin


@pytest.mark.parametrize("n_total,hop,wu,world", [(1000, 256, 16384, 8), (7, 256, 16384, 8), (65536, 256, 16384, 3), (100, 6336, 16384, 4),
                                                  (1 << 20, 256, 16384, 8), (0, 256, 16384, 2)])
def test_plan_shard_covers_the_stream_exactly(n_total, hop, wu, world):
    frames = 0
    for r in range(world):
        s = plan_shard(n_total, hop, wu, r, world)
        assert (s.first_frame, s.n_frames, s.sample_begin, s.sample_end, s.n_lead) == _plan_py(n_total, hop, wu, r, world)
        assert s.first_frame == frames   # contiguous, in order
        frames += s.n_frames
        assert s.sample_end - s.sample_begin == s.n_lead + s.n_frames * hop
        assert s.n_lead == min(max(wu - hop, 0), s.first_frame * hop)   # the whole halo, or all there is before the shard
    assert frames == n_total


def test_plan_shard_rejects_bad_ranks():
    L = _lib.load()
    out = _lib.CShard()
    assert L.pvq_plan_shard(10, 256, 16384, 2, 2, C.byref(out)) == _lib.PVQ_ERR_INVALID_ARG
    assert L.pvq_plan_shard(10, 256, 16384, 0, 0, C.byref(out)) == _lib.PVQ_ERR_INVALID_ARG
    assert L.pvq_plan_shard(10, 256, 16384, 0, 1, None) == _lib.PVQ_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        plan_shard(10, 256, 16384, 3, 2)


def test_multi_needs_devices_and_distinct_handles():
    v = P.Vqt(P.VqtParameters(), device=None)   # host-only plan
    with pytest.raises(P.PvqError) as e:
        P.Vqt.analyze_batch_multi([v], np.zeros(256 * 4, np.float32), 256, 4)
    assert e.value.status == _lib.PVQ_ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("geom,hop", [("bench_48k_252", 256), ("bench_48k_252", 1000), ("default_22k_588", 256)])
def test_k_virtual_shards_equal_one_handle_bit_for_bit(geom, hop):
    import torch
    pp, op = get_geom(geom)
    n_frames, n_lead = 3001, 333
    pcm = white_noise(n_lead + hop * n_frames, 77)
    pcm[5 * hop:9 * hop] *= 40.0   # a loud stretch (the shift branch of power_to_db) and a silent one
    pcm[2000 * hop:2010 * hop] = 0.0
    handles = [P.Vqt.new(pp, 0) for _ in range(8)]
    ref = P.Vqt.analyze_batch_multi(handles[:1], pcm, hop, n_frames, n_lead=n_lead, max_peaks=48)
    # and the single-handle result is the device entry point's
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((n_frames, handles[0].n_bins), device="cuda")
    handles[0].calculate_batch_db_device(d_pcm, hop, n_frames, d_db, n_lead=n_lead)
    torch.cuda.synchronize()
    assert np.array_equal(d_db.cpu().numpy().view(np.uint32), ref[0].view(np.uint32))
    for k in (2, 3, 5, 8):
        got = P.Vqt.analyze_batch_multi(handles[:k], pcm, hop, n_frames, n_lead=n_lead, max_peaks=48)
        for name, a, b in zip(("db", "mask", "count", "center", "size"), ref, got):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, name)
    # dB only (no peak outputs), and errors: a handle twice, mismatched parameters
    only = P.Vqt.analyze_batch_multi(handles[:4], pcm, hop, n_frames, n_lead=n_lead, want_peaks=False)
    assert np.array_equal(only[0].view(np.uint32), ref[0].view(np.uint32)) and only[1] is None
    with pytest.raises(P.PvqError):
        P.Vqt.analyze_batch_multi([handles[0], handles[0]], pcm, hop, n_frames, n_lead=n_lead)
    other = P.Vqt.new(P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 6, 36)), 0)
    with pytest.raises(P.PvqError):
        P.Vqt.analyze_batch_multi([handles[0], other], pcm, hop, n_frames, n_lead=n_lead)


@pytest.mark.gpu
def test_fewer_frames_than_handles_and_repeated_calls():
    """5 frames over 8 handles: three shards are empty and simply do nothing; and the handles' persistent shard buffers (one stream and
    six grow-only device buffers per handle) are reused by a second, larger, then a smaller call with the same bits as one handle."""
    pp, _ = get_geom("bench_48k_252")
    hop = 256
    handles = [P.Vqt.new(pp, 0) for _ in range(8)]
    for n_frames in (5, 700, 64, 5):
        pcm = white_noise(hop * n_frames, 31 + n_frames)
        ref = P.Vqt.analyze_batch_multi(handles[:1], pcm, hop, n_frames, max_peaks=32)
        got = P.Vqt.analyze_batch_multi(handles, pcm, hop, n_frames, max_peaks=32)
        for name, a, b in zip(("db", "mask", "count", "center", "size"), ref, got):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n_frames, name)


@pytest.mark.gpu
def test_a_non_finite_sample_in_exactly_one_shard_names_it():
    """NaN / Inf policy through the multi-device driver: the shard whose frames see the bad sample returns PVQ_ERR_NONFINITE_INPUT and the
    error text names it; a clean call on the same handles afterwards succeeds (every worker reads and clears its own handle's flag)."""
    pp, _ = get_geom("bench_48k_252")
    hop, n_frames, k = 256, 4000, 4
    handles = [P.Vqt.new(pp, 0) for _ in range(k)]
    pcm = white_noise(hop * n_frames, 5)
    clean = P.Vqt.analyze_batch_multi(handles, pcm, hop, n_frames, max_peaks=16)
    bad = pcm.copy()
    bad[2600 * hop + 17] = np.inf   # frames 2600 .. 2663 see it: shard 2 of 4 (frames 2000 .. 2999) only
    with pytest.raises(P.PvqError) as e:
        P.Vqt.analyze_batch_multi(handles, bad, hop, n_frames, max_peaks=16)
    assert e.value.status == _lib.PVQ_ERR_NONFINITE_INPUT and "shard 2" in str(e.value)
    again = P.Vqt.analyze_batch_multi(handles, pcm, hop, n_frames, max_peaks=16)
    for a, b in zip(clean, again):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
