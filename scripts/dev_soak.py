"""Determinism soak over the test geometries (developer tool): the same batch N times, every output compared bit for bit."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
from helpers import GEOMS, get_geom
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for name in GEOMS:
    pp, _ = get_geom(name)
    for prec in (0, 1):
        v = P.Vqt(pp, 0); v.set_algo(2); v.set_gemm_precision(prec)
        hop = 128 if pp.sr > 90000 else 256
        nf = 20000
        gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
        d_pcm = (torch.rand(hop * nf + 777, device="cuda", generator=gen) - 0.5) * 0.5
        words = (v.n_bins + 31) // 32
        ref = None
        for it in range(runs):
            d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
            d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
            d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
            v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=777, d_out_cplx=d_cx)
            v.analyze_batch_device(d_db, nf, d_mask, d_cnt, d_c, d_s, 64)
            torch.cuda.synchronize()
            cur = (d_db, d_cx, d_mask, d_cnt, d_c, d_s)
            if ref is None:
                ref = cur
            else:
                for k, (a, b) in enumerate(zip(ref, cur)):
                    if not torch.equal(a, b):
                        bad += 1
                        print(f"MISMATCH {name} prec {prec} run {it} output {k}: {int((a != b).sum())} entries", flush=True)
        print(f"{name:18s} prec {prec}: {runs} runs ok" if bad == 0 else f"{name} prec {prec}: done ({bad} mismatches so far)", flush=True)
print("SOAK", "CLEAN" if bad == 0 else f"FAILED {bad}")
