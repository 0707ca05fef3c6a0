"""NaN / Inf input policy on the GPU (include/pvq.h): a non-finite sample inside one of a frame's windows
is reported — PVQ_ERR_NONFINITE_INPUT from the synchronous host-buffer entry points, pvq_vqt_input_status for the
asynchronous device-pointer ones — instead of silently turning into the A_MIN floor.  The reference never lets such
samples reach the transform (audio_desktop.rs:102-105) and would panic on their NaNs (peak_detection.rs:145)."""
import numpy as np
import pytest

import pitchvis_amd as P
from pitchvis_amd import _lib
from helpers import get_geom, white_noise

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("algo", [P.ALGO_FFT, P.ALGO_BLOCKDFT])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_nonfinite_sample_is_reported_and_the_flag_clears(algo, bad):
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    v.set_algo(algo)
    hop, nf = 256, 300
    clean = white_noise(hop * nf, 5)
    ref = v.calculate_batch_db(clean, hop, nf)               # finite input: no error
    dirty = clean.copy()
    dirty[hop * 150 + 7] = bad
    with pytest.raises(P.PvqError) as e:
        v.calculate_batch_db(dirty, hop, nf)
    assert e.value.status == _lib.PVQ_ERR_NONFINITE_INPUT
    again = v.calculate_batch_db(clean, hop, nf)             # the flag was cleared by the failing call
    assert np.array_equal(again, ref)
    # asynchronous entry point: nothing is raised by the launch, input_status reports once
    d_pcm = torch.from_numpy(dirty).cuda()
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
    with pytest.raises(P.PvqError) as e:
        v.input_status()
    assert e.value.status == _lib.PVQ_ERR_NONFINITE_INPUT
    v.input_status()                                         # cleared
    # frames whose windows do not contain the sample are untouched
    got = d_db.cpu().numpy()
    first_hit = 150                                          # the frame that first sees sample 150 * hop + 7
    assert np.array_equal(got[:first_hit], ref[:first_hit])
    last_hit = first_hit + v.window_union // hop + 1
    assert np.array_equal(got[last_hit + 1:], ref[last_hit + 1:])


def test_nonfinite_history_outside_every_window_is_not_an_input():
    """n_lead samples older than the window union are never read: a NaN there must not raise the flag."""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, nf, n_lead = 256, 128, 40000
    pcm = white_noise(n_lead + hop * nf, 6)
    pcm[100] = np.nan                                        # 40 000 - 100 > window union (16 384)
    v.calculate_batch_db(pcm, hop, nf, n_lead=n_lead)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead)
    v.input_status()
    assert torch.isfinite(d_db).all()


def test_instant_call_reports_it_too():
    pp, op = get_geom("default_22k_588")
    v = P.Vqt.new(pp, 0)
    x = white_noise(op.n_fft, 9)
    v.calculate_vqt_instant_in_db(x)
    x[-5] = np.inf
    with pytest.raises(P.PvqError) as e:
        v.calculate_vqt_instant_in_db(x)
    assert e.value.status == _lib.PVQ_ERR_NONFINITE_INPUT


def test_a_synchronous_call_answers_for_its_own_input_only():
    """The flag is per handle: an asynchronous call that raised it and was never polled must not make the next synchronous
    call fail on clean samples (include/pvq.h: the synchronous entry points clear what they did not raise)."""
    pp, _ = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    hop, nf = 256, 200
    clean = white_noise(hop * nf, 15)
    dirty = clean.copy()
    dirty[hop * 50] = np.nan
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    v.calculate_batch_db_device(torch.from_numpy(dirty).cuda(), hop, nf, d_db)   # raises the flag, nobody polls
    ref = v.calculate_batch_db(clean, hop, nf)                                    # clean input: PVQ_OK
    assert np.isfinite(ref).all()
    v.input_status()                                                               # and nothing is left behind


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_single_frame_call_checks_its_window_union_on_the_host(bad):
    """pvq_vqt_calculate_instant_db (the reference's call shape) stages only the window union of its n_fft samples and looks at them
    while it does: a non-finite sample among them is PVQ_ERR_NONFINITE_INPUT before anything is launched — no flag is left on the
    device for a later call to trip over —, one before them (never read) is not an input"""
    pp, op = get_geom("bench_48k_252")
    v = P.Vqt.new(pp, 0)
    x = white_noise(pp.n_fft, 9)
    ref = v.calculate_vqt_instant_in_db(x)
    wu = v.window_union
    for at in (pp.n_fft - 1, pp.n_fft - wu, pp.n_fft - wu // 2):
        y = x.copy(); y[at] = bad
        with pytest.raises(P.PvqError) as e:
            v.calculate_vqt_instant_in_db(y)
        assert e.value.status == _lib.PVQ_ERR_NONFINITE_INPUT
        v.input_status()                                     # nothing was raised on the device
    y = x.copy(); y[pp.n_fft - wu - 1] = bad; y[0] = bad     # older than every window
    assert np.array_equal(v.calculate_vqt_instant_in_db(y), ref)
    assert np.array_equal(v.calculate_vqt_instant_in_db(x), ref)
    # ... and the call is the same frame a batch of one gives
    d = torch.from_numpy(x).cuda(); d_db = torch.empty((1, v.n_bins), device="cuda")
    v.set_algo(P.ALGO_FFT)
    v.calculate_batch_db_device(d, pp.n_fft, 1, d_db)
    assert np.array_equal(d_db.cpu().numpy()[0], ref)
