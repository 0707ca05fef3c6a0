import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
for octv, bpo in ((8, 36), (7, 36)):
    pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, octv, bpo))
    v = P.Vqt(pp, 0)
    hop, nf = 256, 65536
    d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
    d_db = torch.empty((nf, v.n_bins), device="cuda")
    words = (v.n_bins+31)//32
    d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
    d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
    fn = lambda: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64)
    for _ in range(3): fn()
    torch.cuda.synchronize(); v.set_profiling(True); t = time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.time()-t)/10
    km = v.last_kernel_ms()
    print(v.n_bins, f"{dt*1e3:.3f} ms/step {nf/dt/1e6:.1f} Mf/s", {k: round(x*1e3) for k, x in km.items()})
