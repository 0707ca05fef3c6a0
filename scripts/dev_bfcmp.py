"""Developer tool: per-bin difference of the complex coefficients between the fp32 and split-bf16 GEMM forms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
from helpers import get_geom
name = sys.argv[1] if len(sys.argv) > 1 else "hires_96k_360"
pp, _ = get_geom(name)
hop = 128 if pp.sr > 90000 else 256
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rng = np.random.default_rng(1)
pcm = ((rng.random(hop * nf) - 0.5) * 0.5).astype(np.float32)
res = []
for prec in (0, 1):
    v = P.Vqt(pp, 0); v.set_algo(2); v.set_gemm_precision(prec)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    d_cx = torch.empty((nf, v.n_bins, 2), device="cuda")
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, d_out_cplx=d_cx)
    torch.cuda.synchronize()
    c = d_cx.cpu().numpy()
    res.append(c[..., 0] + 1j * c[..., 1])
d = np.abs(res[0] - res[1])
sc = np.abs(res[0]).max()
print("scale", sc, "max diff", d.max() / sc)
perbin = d.max(axis=0) / sc
perframe = d.max(axis=1) / sc
print("bins with diff > 1e-5:", np.nonzero(perbin > 1e-5)[0])
bad = np.nonzero(perframe > 1e-5)[0]
print("frames with diff > 1e-5:", len(bad), bad[:40], bad[-10:] if len(bad) else "")
