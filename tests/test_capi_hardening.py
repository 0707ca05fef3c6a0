"""The C ABI keeps its promises without a GPU (include/pvq.h: "no exceptions cross the ABI"): where the reference
panics during kernel construction (vqt.rs:785-792) the call returns PVQ_ERR_INVALID_ARG with the reference's text
instead of aborting the host process; std::bad_alloc / std::length_error / anything else thrown inside the library
is caught at the boundary and becomes PVQ_ERR_INTERNAL."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import pytest

import pitchvis_amd as P
from pitchvis_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_panics_become_invalid_arg_with_the_reference_text():
    # gamma < 0 makes a filter window overrun its group window: `assert!(filter_begin + len <= scaled_n_fft, "filter
    # window must end before the end of its group window")` in the reference (vqt.rs:788-792); round 1 abort()ed here
    with pytest.raises(P.PvqError) as e:
        P.Vqt(P.VqtParameters(48000.0, 4096, P.VqtRange(53.79554329358336, 5, 84), 0.9772649201225388,
                              3.514094482643492, -4.711600280106042), device=None)
    assert e.value.status == _lib.PVQ_ERR_INVALID_ARG
    assert "filter window must end before the end of its group window" in str(e.value)
    # the process is alive and the library still works
    v = P.Vqt(P.VqtParameters(), device=None)
    assert v.n_bins == 588


def test_status_strings_cover_every_status():
    L = _lib.load()
    names = [L.pvq_status_string(i).decode() for i in range(10)]
    assert len(set(names)) == 10 and "unknown" not in names
    assert "non-finite" in names[_lib.PVQ_ERR_NONFINITE_INPUT] and "internal" in names[_lib.PVQ_ERR_INTERNAL]
    assert L.pvq_abi_version() == 4


@pytest.mark.parametrize("what,text", [("bad_alloc", "out of host memory"), ("length_error", "vector"), ("int", "unknown exception")])
def test_exceptions_stop_at_the_abi(what, text):
    """PVQ_TEST_THROW makes pvq_vqt_create of the DEVELOPER library (libpvq_dev.so; the product build has no such hook) throw inside
    its guarded body (a child process keeps the environment of this one clean).  Without the barrier the child would die in std::terminate."""
    code = textwrap.dedent(f"""
        import ctypes as C, sys
        sys.path.insert(0, {ROOT!r})
        from pitchvis_amd import _lib
        L = _lib.load()
        p = _lib.CParams(); L.pvq_vqt_default_params(C.byref(p))
        h = C.c_void_p(); err = (C.c_float * 2)()
        st = L.pvq_vqt_create(C.byref(p), -1, C.byref(h), err)
        print("STATUS", st, "|", L.pvq_last_error().decode(), "|", bool(h.value))
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PVQ_TEST_THROW=what, PVQ_DEV_LIB="1"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("STATUS")][0]
    assert line.startswith(f"STATUS {_lib.PVQ_ERR_INTERNAL} |") and text in line and line.endswith("False")


def test_null_and_range_arguments_return_statuses():
    L = _lib.load()
    assert L.pvq_vqt_input_status(None, None) == _lib.PVQ_ERR_INVALID_ARG
    assert L.pvq_vqt_last_gemm_flop(None) == 0.0 and L.pvq_vqt_last_sclk_mhz(None) == 0.0
    v = P.Vqt(P.VqtParameters(), device=None)
    v.input_status()   # host-only handle: nothing to check, PVQ_OK
    info = (C.c_uint32 * 5)()
    assert L.pvq_vqt_group_info(v._h, 99, info) == _lib.PVQ_ERR_INVALID_ARG
