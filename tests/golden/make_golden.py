"""Generates the committed golden fixtures from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself ships no golden vectors and cannot
be built here (Rust), so these pin the ORACLE's outputs — a regression anchor for
both the oracle and the GPU path — not the Rust binary's.  Inputs are stored with the outputs."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
from helpers import GEOMS, get_geom, three_regime, white_noise, sine_sweep  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def kernel_digest(ov):
    h = hashlib.sha256()
    for g in range(ov.n_groups):
        gi = ov.group_info(g)
        h.update(np.array(gi["window"], np.uint32).tobytes())
        for neg in (False, True):
            rp, ci, va = ov.group_csr(g, neg)
            h.update(rp.tobytes()); h.update(ci.tobytes()); h.update(va.tobytes())
    return h.hexdigest()


def main():
    digests = {}
    for name in GEOMS:
        _, op = get_geom(name)
        ov = O.OracleVqt(op)
        digests[name] = kernel_digest(ov)
    with open(os.path.join(OUT, "kernel_digests.txt"), "w") as f:
        for k, v in digests.items():
            f.write(f"{k} {v}\n")

    # frames: bench geometry, hop 256, with history (n_lead) so the windows are full
    _, op = get_geom("bench_48k_252")
    ov = O.OracleVqt(op)
    hop, nf, n_lead = 256, 24, 16384
    cases = {
        "noise": white_noise(n_lead + hop * nf, 0x5EED0001),
        "sweep": sine_sweep(n_lead + hop * nf, op.sr),
        "regimes": three_regime(n_lead + hop * nf, op.sr, 4),
    }
    pack = {}
    for cname, pcm in cases.items():
        db, cx = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
        peaks = np.zeros((nf, (ov.n_bins + 31) // 32), np.uint32)
        for f in range(nf):
            for p in O.find_peaks_split(db[f], op.buckets_per_octave):
                peaks[f, p // 32] |= np.uint32(1 << (p % 32))
        pack[f"{cname}_pcm"] = pcm
        pack[f"{cname}_db"] = db
        pack[f"{cname}_cplx"] = cx
        pack[f"{cname}_peakmask"] = peaks
    pack["hop"] = np.array(hop); pack["n_frames"] = np.array(nf); pack["n_lead"] = np.array(n_lead)
    np.savez_compressed(os.path.join(OUT, "bench_48k_252_frames.npz"), **pack)

    # the reference default geometry: stream start (zeros before the stream), odd hop
    _, op = get_geom("default_22k_588")
    ov = O.OracleVqt(op)
    hop, nf = 441, 16
    pcm = white_noise(hop * nf, 0x5EED0002, amp=0.5)
    db = ov.calculate_batch(pcm, hop, nf)
    np.savez_compressed(os.path.join(OUT, "default_22k_588_frames.npz"), pcm=pcm, db=db, hop=np.array(hop),
                        n_frames=np.array(nf))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
