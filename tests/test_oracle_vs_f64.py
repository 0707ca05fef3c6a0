"""Cross-check of the f32 C oracle against the independent float64 NumPy model."""
import numpy as np
import pytest

import oracle as O
from oracle import model_f64 as MF
from helpers import GEOMS, get_geom, white_noise


@pytest.mark.parametrize("name", list(GEOMS))
def test_integer_decisions_and_windows_agree(name):
    _, op = get_geom(name)
    ov = O.OracleVqt(op)
    m = MF.from_oracle_params(op, pattern_from=ov)
    freq, wl, M, mw = ov.filter_params()
    assert (M == m.M).all() and (mw == m.minwin).all()
    assert np.allclose(freq, m.freq, rtol=2e-6) and np.allclose(wl, m.wl, rtol=2e-5)
    assert ov.n_groups == len(m.groups)
    for g in range(ov.n_groups):
        assert ov.group_info(g)["window"] == m.groups[g]["window"]
    assert abs(ov.delay - m.delay) < 1e-6


@pytest.mark.parametrize("name", ["default_22k_588", "bench_48k_252"])
def test_kernel_values_close_to_exact_math(name):
    """With the oracle's sparsity pattern, coefficient values agree with exact (f64) math to the
    level the reference's f32 phase evaluation allows (~1e-4 of the group's largest coefficient)."""
    _, op = get_geom(name)
    ov = O.OracleVqt(op)
    m = MF.from_oracle_params(op, pattern_from=ov)
    for g in range(ov.n_groups):
        rp, ci, va = ov.group_csr(g)
        K = m.groups[g]["K"]
        dense = np.zeros_like(K)
        for r in range(len(rp) - 1):
            dense[r, ci[rp[r]:rp[r + 1]]] = va[rp[r]:rp[r + 1]]
        assert np.abs(dense - K).max() / np.abs(K).max() < 3e-4


@pytest.mark.parametrize("name", ["default_22k_588", "bench_48k_252", "hires_96k_360"])
def test_frames_match_f64_given_the_kernel(name):
    """rFFT + sparse products in f32 vs exact: error <= 1e-6 of the frame's largest magnitude."""
    _, op = get_geom(name)
    ov = O.OracleVqt(op)
    m = MF.from_oracle_params(op, values_from=ov)
    x = white_noise(op.n_fft, 11)
    c32 = ov.calculate_vqt_instant_complex(x)
    c64 = m.frame_complex(x)
    assert np.abs(c32 - c64).max() / np.abs(c64).max() < 1e-6
    assert np.abs(ov.calculate_vqt_instant_in_db(x) - m.frame_db(x)).max() < 1e-3


def test_own_pattern_model_agrees_loosely():
    """Fully independent model (own sparsity decisions): a flipped borderline coefficient moves an
    output by < 1e-3 of the frame maximum."""
    _, op = get_geom("bench_48k_252")
    ov = O.OracleVqt(op)
    m = MF.from_oracle_params(op)
    x = white_noise(op.n_fft, 12)
    c32 = ov.calculate_vqt_instant_complex(x)
    c64 = m.frame_complex(x)
    assert np.abs(np.abs(c32) - np.abs(c64)).max() / np.abs(c64).max() < 1e-3
