"""Analytic known answers and edge cases for the CPU oracle."""
import numpy as np
import pytest

import oracle as O
from oracle import model_f64 as MF
from helpers import get_geom, three_regime


@pytest.mark.parametrize("geom,bin_,expect", [("default_22k_588", 300, 26.287), ("bench_48k_252", 130, 29.666)])
def test_on_centre_sine_kat(geom, bin_, expect):
    """|X_k| = sqrt(sr)*a/2 for an on-centre sine of amplitude a = 1/12 (L1 normalisation
    vqt.rs:802-805, kernel_gain vqt.rs:646, REF_POWER vqt.rs:923); far bins clamp to 0."""
    _, op = get_geom(geom)
    ov = O.OracleVqt(op)
    freq = ov.filter_params()[0]
    db = ov.calculate_vqt_instant_in_db(O.test_create_sines(op, [freq[bin_]]))
    analytic = 20 * np.log10(np.sqrt(op.sr) * (1 / 12) / 2) - 10 * np.log10(0.09)
    assert abs(analytic - expect) < 1e-3
    assert db.argmax() == bin_
    assert abs(db.max() - analytic) < 0.01
    assert (db[np.abs(np.arange(db.size) - bin_) > 40] == 0).all()


def test_power_to_db_branches():
    """vqt.rs:922-954: silence -> all zero; clip branch; shift branch"""
    n = 64
    z = np.zeros(n, np.complex64)
    assert (O.power_to_db(z) == 0).all()
    # clip branch: some bin below 0 dB (|z|^2 < 0.09)
    z = (np.linspace(0.01, 3.0, n) + 0j).astype(np.complex64)
    d = O.power_to_db(z)
    raw = 10 * np.log10(np.abs(z.astype(np.complex128)) ** 2) - 10 * np.log10(0.09)
    assert np.allclose(d, np.maximum(raw, 0.0), atol=1e-4)
    # shift branch: every bin > 0 dB -> minimum becomes exactly 0
    z = (np.linspace(1.0, 30.0, n) + 0j).astype(np.complex64)
    d = O.power_to_db(z)
    raw = 10 * np.log10(np.abs(z.astype(np.complex128)) ** 2) - 10 * np.log10(0.09)
    assert d.min() == 0.0 and np.allclose(d, raw - raw.min(), atol=1e-4)
    # 60 dB floor
    z = np.array([1e-5, 1e3], np.complex64)
    d = O.power_to_db(z)
    assert abs(d[1] - d[0] - 60.0) < 1e-3 or d[0] == 0.0


def test_three_regime_input_hits_both_db_branches():
    _, op = get_geom("bench_48k_252")
    ov = O.OracleVqt(op)
    hop, nf = 2048, 48
    pcm = three_regime(hop * nf, op.sr, 3)
    db, cx = ov.calculate_batch(pcm, hop, nf, want_complex=True)
    raw = 10 * np.log10(np.maximum(np.abs(cx.astype(np.complex128)) ** 2, 1e-12)) - 10 * np.log10(0.09)
    silent = (np.abs(cx) == 0).all(axis=1)
    shift = (np.maximum(raw.min(axis=1), raw.max(axis=1) - 60) > 0)
    assert silent.sum() >= 3 and shift.sum() >= 3 and ((~silent) & (~shift)).sum() >= 3
    assert (db[silent] == 0).all()
    assert (db[shift].min(axis=1) == 0).all()


def test_dc_and_impulse_edges():
    _, op = get_geom("bench_48k_252")
    ov = O.OracleVqt(op)
    m = MF.from_oracle_params(op, values_from=ov)
    for x in (np.full(op.n_fft, 0.5, np.float32), np.eye(1, op.n_fft, op.n_fft - 3000, dtype=np.float32)[0]):
        c32 = ov.calculate_vqt_instant_complex(x)
        c64 = m.frame_complex(x)
        scale = max(np.abs(c64).max(), 1e-9)
        assert np.abs(c32 - c64).max() / scale < 2e-5 or np.abs(c64).max() < 1e-6


def test_batch_framing_matches_instant_calls():
    """frame f of the hop stream == calculate_vqt_instant_in_db on the ring buffer after hop f"""
    _, op = get_geom("serial_22k_180")
    ov = O.OracleVqt(op)
    hop, nf = 441, 12  # deliberately not a power of two
    rng = np.random.default_rng(5)
    pcm = (rng.random(hop * nf + 1234, dtype=np.float32) - 0.5).astype(np.float32)
    n_lead = 1234
    db = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead)
    ring = np.zeros(op.n_fft, np.float32)
    ring[-n_lead:] = pcm[:n_lead]
    for f in range(nf):
        chunk = pcm[n_lead + f * hop: n_lead + (f + 1) * hop]
        ring = np.concatenate([ring[hop:], chunk])  # audio_desktop.rs:113-115
        assert np.array_equal(db[f], ov.calculate_vqt_instant_in_db(ring))


def test_errors():
    with pytest.raises(O.OracleVqtError) as e:
        O.OracleVqt(O.OracleParams(sr=96000.0, min_freq=55.0, octaves=10, buckets_per_octave=36))
    assert e.value.code == 1 and abs(e.value.a - 55246.0) < 1.0 and e.value.b == 48000.0
    with pytest.raises(O.OracleVqtError) as e:
        O.OracleVqt(O.OracleParams(quality=30.0))
    assert e.value.code == 2 and e.value.b == 32768.0
