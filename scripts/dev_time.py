"""Kernel timing variants on the bench workload (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
algo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0); v.set_algo(algo)
if len(sys.argv) > 2: v.set_gemm_precision(int(sys.argv[2]))
hop, nf = 256, 65536
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
words = (v.n_bins+31)//32
d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
def run(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    v.set_profiling(True)
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.time()-t)/n
    km = v.last_kernel_ms(); kn = v.last_kernel_launches()
    v.set_profiling(False)
    print(f"{name:28s} {dt*1e3:7.3f} ms/step  {nf/dt/1e6:6.2f} Mf/s  " + "  ".join(f"{k}={km[k]*1e3:.1f}us x{kn[k]//n}" for k in km))
run("vqt only", lambda: v.calculate_batch_db_device(d_pcm, hop, nf, d_db))
if "once" in sys.argv: sys.exit(0)
if "withpeaks" in sys.argv:
    run("vqt + mask/count/continuous", lambda: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64), n=3)
    sys.exit(0)
run("vqt + mask/count", lambda: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt))
run("vqt + mask/count/continuous", lambda: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64))
run("standalone peaks full", lambda: v.analyze_batch_device(d_db, nf, d_mask, d_cnt, d_c, d_s, 64))
run("standalone peaks mask only", lambda: v.analyze_batch_device(d_db, nf, d_mask, d_cnt))
