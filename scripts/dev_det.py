"""Determinism probe (developer tool): run the same batch several times, report where outputs differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0); v.set_algo(2); v.set_gemm_precision(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
hop, nf = 256, 65536
gen = torch.Generator(device="cuda"); gen.manual_seed(0x5EED0001)
d_pcm = (torch.rand(hop * nf, device="cuda", generator=gen) - 0.5) * 0.5
outs = []
for it in range(4):
    d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, d_out_cplx=d_cx); torch.cuda.synchronize()
    outs.append((d_db, d_cx))
ref = P.Vqt(pp, 0); ref.set_algo(2); ref.set_gemm_precision(0)
r_db = torch.empty((nf, v.n_bins), device="cuda"); r_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
ref.calculate_batch_db_device(d_pcm, hop, nf, r_db, d_out_cplx=r_cx); torch.cuda.synchronize()
for it in range(4):
    e = (outs[it][1] - r_cx).abs().amax(dim=(1, 2)) / r_cx.abs().amax(dim=(1, 2))
    bad = torch.nonzero(e > 1e-4).flatten().cpu().numpy()
    print(f"run {it} vs fp32 handle: max rel err {float(e.max()):.3g}, frames off by > 1e-4: {bad[:10]}", flush=True)
for it in range(1, 4):
    dc = (outs[it][1] != outs[0][1]).any(dim=2)
    dd = outs[it][0] != outs[0][0]
    fr = torch.nonzero(dc.any(dim=1)).flatten().cpu().numpy()
    bins = torch.nonzero(dc.any(dim=0)).flatten().cpu().numpy()
    print(f"run {it}: cx differs in {int(dc.sum())} entries, {len(fr)} frames {fr[:12]}, bins {bins[:20]}; db differs in {int(dd.sum())} entries", flush=True)
    if len(fr):
        f = int(fr[0]); b = int(torch.nonzero(dc[f]).flatten()[0])
        print("   e.g.", outs[0][1][f, b].cpu().numpy(), outs[it][1][f, b].cpu().numpy())
