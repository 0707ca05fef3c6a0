"""Throughput of the whole path at the test geometries (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
from helpers import GEOMS, get_geom
for name in GEOMS:
    pp, _ = get_geom(name)
    v = P.Vqt(pp, 0)
    hop = 128 if pp.sr > 90000 else 256
    nf = 32768
    d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    words = (v.n_bins+31)//32
    d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
    d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
    fn = lambda: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64)
    for _ in range(2): fn()
    torch.cuda.synchronize(); v.set_profiling(True); t = time.time()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt = (time.time()-t)/5
    km = v.last_kernel_ms(); kn = v.last_kernel_launches()
    print(f"{name:18s} bins {v.n_bins:4d} hop {hop}: {dt*1e3:8.3f} ms per {nf} frames = {nf/dt/1e6:6.1f} Mf/s  algo {v.last_algo()}", {k: round(x*1e3) for k, x in km.items()}, flush=True)
