"""CPU-side checks of the product library: the C ABI loads and exports what include/pvq.h
declares, host kernel construction is bit-identical to the oracle, errors mirror VqtError, and
compute refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from pitchvis_amd import _lib
from helpers import GEOMS, get_geom

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "pvq.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pvq_[a-z0-9_]+)\s*\(", hdr)))
    L = _lib.load()
    assert declared, "no declarations parsed"
    assert sorted(_lib.EXPORTS) == declared
    for name in declared:
        assert hasattr(L, name), f"libpvq.so lacks {name}"
    assert L.pvq_abi_version() == 4


def test_no_oracle_dependency_in_product():
    """the product path must not import, link or call anything under oracle/"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pitchvis_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"import\s+oracle|from\s+oracle|oracle/|liboracle|pvq_oracle|orc_", txt), f"{f} uses the oracle"
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out


def test_default_params_match_reference_constants():
    p = P.VqtParameters.default()  # vqt.rs:180-214
    assert (p.sr, p.n_fft, p.range.min_freq, p.range.octaves, p.range.buckets_per_octave) == (22050.0, 32768, 55.0, 7, 84)
    assert abs(p.sparsity_quantile - 0.999) < 1e-7 and abs(p.quality - 1.6) < 1e-7 and abs(p.gamma - 7.68) < 1e-6
    assert p.range.n_buckets() == 7 * 84  # vqt.rs:224-237 doc-test


@pytest.mark.parametrize("name", list(GEOMS))
def test_host_kernel_bit_identical_to_oracle(name):
    pp, op = get_geom(name)
    v = P.Vqt.new(pp, device=None)
    ov = O.OracleVqt(op)
    k = v.kernel()
    assert len(k.window_groups) == ov.n_groups
    assert v.delay == ov.delay and v.n_bins == ov.n_bins
    for g, wg in enumerate(k.window_groups):
        gi = ov.group_info(g)
        assert wg.window == gi["window"] and wg.filter_bank.rows() == gi["rows"]
        rp, ci, va = ov.group_csr(g)
        assert np.array_equal(rp, wg.filter_bank.indptr) and np.array_equal(ci, wg.filter_bank.indices)
        assert np.array_equal(va.view(np.uint32), wg.filter_bank.data.view(np.uint32))  # bit-exact
        if gi["neg_nnz"]:
            rp, ci, va = ov.group_csr(g, neg=True)
            assert np.array_equal(ci, wg.negative_filter_bank.indices)
            assert np.array_equal(va.view(np.uint32), wg.negative_filter_bank.data.view(np.uint32))
        else:
            assert wg.negative_filter_bank is None  # vqt.rs:751
    f1, w1, m1, mw1 = v.filter_params()
    f2, w2, m2, mw2 = ov.filter_params()
    assert np.array_equal(f1, f2) and np.array_equal(w1, w2) and np.array_equal(m1, m2) and np.array_equal(mw1, mw2)


def test_reference_default_kernel_shape():
    """vqt.rs:133-134 / VQT_REVIEW.md:364,369: 4 real FFTs 8192/4096/2048/1024, 379 conj-part nnz"""
    v = P.Vqt.new(P.VqtParameters.default(), device=None)
    k = v.kernel()
    assert [g.window_size() for g in k.window_groups] == [8192, 4096, 2048, 1024]
    assert sum(g.negative_filter_bank.nnz() for g in k.window_groups if g.negative_filter_bank) == 379
    total = sum(g.filter_bank.nnz() for g in k.window_groups) + 379
    assert 17000 < total < 19000  # "~18 k non-zeros"
    assert int(v.delay * 1000) == 98  # vqt.rs:1078-1085, VQT_REVIEW.md:363
    assert v.window_union == 8192


def test_errors_mirror_vqt_error():
    with pytest.raises(P.AboveNyquist) as e:  # vqt.rs:518-528; SURVEY.md §0 (96 kHz, 10 oct above 55 Hz)
        P.Vqt.new(P.VqtParameters(sr=96000.0, range=P.VqtRange(55.0, 10, 36)), device=None)
    assert abs(e.value.highest_frequency - 55246.0) < 1.0 and e.value.nyquist_frequency == 48000.0
    assert "exceeds the Nyquist frequency" in str(e.value)
    with pytest.raises(P.WindowExceedsNFft) as e:  # vqt.rs:567-573
        P.Vqt.new(P.VqtParameters(quality=30.0), device=None)
    assert e.value.n_fft == 32768 and "exceeds n_fft" in str(e.value)
    with pytest.raises(P.PvqError):
        P.Vqt.new(P.VqtParameters(n_fft=30000), device=None)


def test_compute_without_device_fails_loudly():
    v = P.Vqt.new(P.VqtParameters.default(), device=None)
    with pytest.raises(P.PvqError) as e:
        v.calculate_vqt_instant_in_db(np.zeros(32768, np.float32))
    assert e.value.status == _lib.PVQ_ERR_NO_DEVICE
    with pytest.raises(P.PvqError):
        v.calculate_batch_db(np.zeros(4096, np.float32), 256)
    with pytest.raises(P.PvqError):
        v.analyze_batch(np.zeros((1, v.n_bins), np.float32))
    with pytest.raises(AssertionError):  # vqt.rs:867-871 panics on a wrong length
        v.calculate_vqt_instant_in_db(np.zeros(100, np.float32))
