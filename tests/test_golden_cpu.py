"""Oracle (and the product's host-side kernel construction) against the committed golden fixtures."""
import hashlib
import os

import numpy as np
import pytest

import oracle as O
import pitchvis_amd as P
from helpers import GEOMS, get_geom, mask_to_indices

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _digest_product(v):
    h = hashlib.sha256()
    for wg in v.kernel().window_groups:
        h.update(np.array(wg.window, np.uint32).tobytes())
        for m in (wg.filter_bank, wg.negative_filter_bank):
            if m is None:
                rows = wg.filter_bank.rows()
                h.update(np.zeros(rows + 1, np.uint32).tobytes()); h.update(b""); h.update(b"")
            else:
                h.update(m.indptr.tobytes()); h.update(m.indices.tobytes()); h.update(m.data.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("name", list(GEOMS))
def test_kernel_digests(name):
    want = dict(l.split() for l in open(os.path.join(G, "kernel_digests.txt")))
    pp, op = get_geom(name)
    ov = O.OracleVqt(op)
    h = hashlib.sha256()
    for g in range(ov.n_groups):
        h.update(np.array(ov.group_info(g)["window"], np.uint32).tobytes())
        for neg in (False, True):
            rp, ci, va = ov.group_csr(g, neg)
            h.update(rp.tobytes()); h.update(ci.tobytes()); h.update(va.tobytes())
    assert h.hexdigest() == want[name]
    assert _digest_product(P.Vqt.new(pp, device=None)) == want[name]


def test_bench_frames_golden():
    z = np.load(os.path.join(G, "bench_48k_252_frames.npz"))
    _, op = get_geom("bench_48k_252")
    ov = O.OracleVqt(op)
    hop, nf, n_lead = int(z["hop"]), int(z["n_frames"]), int(z["n_lead"])
    for case in ("noise", "sweep", "regimes"):
        db, cx = ov.calculate_batch(z[f"{case}_pcm"], hop, nf, n_lead=n_lead, want_complex=True)
        assert np.array_equal(db, z[f"{case}_db"]) and np.array_equal(cx, z[f"{case}_cplx"])
        for f in range(nf):
            assert np.array_equal(O.find_peaks_split(db[f], 36), mask_to_indices(z[f"{case}_peakmask"][f], 252))


def test_default_frames_golden():
    z = np.load(os.path.join(G, "default_22k_588_frames.npz"))
    _, op = get_geom("default_22k_588")
    ov = O.OracleVqt(op)
    db = ov.calculate_batch(z["pcm"], int(z["hop"]), int(z["n_frames"]))
    assert np.array_equal(db, z["db"])
