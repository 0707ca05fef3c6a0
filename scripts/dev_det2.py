import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
hop, nf = 256, 65536
gen = torch.Generator(device="cuda"); gen.manual_seed(0x5EED0001)
d_pcm = (torch.rand(hop * nf, device="cuda", generator=gen) - 0.5) * 0.5
ref = P.Vqt(pp, 0); ref.set_algo(2); ref.set_gemm_precision(0)
r_db = torch.empty((nf, ref.n_bins), device="cuda"); r_cx = torch.empty((nf, ref.n_bins, 2), device="cuda")
ref.calculate_batch_db_device(d_pcm, hop, nf, r_db, d_out_cplx=r_cx); torch.cuda.synchronize()
v = P.Vqt(pp, 0); v.set_algo(2); v.set_gemm_precision(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, d_out_cplx=d_cx); torch.cuda.synchronize()
    e = (d_cx - r_cx).abs().amax(dim=2) / r_cx.abs().amax(dim=(1, 2))[:, None]
    bad = torch.nonzero((e > 1e-4).any(dim=1)).flatten().cpu().numpy()
    bad = bad[bad > 1]
    for f in bad:
        bb = torch.nonzero(e[f] > 1e-4).flatten().cpu().numpy()
        print(f"run {it}: frame {f} (mod 193 = {f % 193}, mod 64 = {f % 64}) bins {bb.min()}..{bb.max()} ({len(bb)}), max rel err {float(e[f].max()):.3g}", flush=True)
print("done")
