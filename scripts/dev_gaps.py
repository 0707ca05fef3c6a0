"""Step time with and without the per-kernel HIP events of profiling mode (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf = 256, 65536
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
words = (v.n_bins+31)//32
d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
def step(): v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64)
for prof in (False, True, False, True):
    v.set_profiling(prof)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    km = v.last_kernel_ms() if prof else {}
    print(f"profiling {prof}: {dt*1e3:.4f} ms/step", {k: round(x*1e3, 1) for k, x in km.items()})
