"""Developer tool: one geometry, one batch shape, n launches (product or developer library) — for chasing a fault or a wrong result.
usage: dev_repro.py OCTAVES N_FRAMES N_LEAD(-1: a full window) [LAUNCHES] [analyze]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P
octaves = int(sys.argv[1]); nf = int(sys.argv[2]); n_lead = int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, octaves, 36))
v = P.Vqt.new(pp, 0)
hop = 256
if n_lead < 0: n_lead = v.window_union - hop
d_pcm = (torch.rand(n_lead + hop * nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
words = (v.n_bins + 31) // 32
d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
if "prof2" in sys.argv: v.set_profiling(2)
for i in range(n):
    if "analyze" in sys.argv:
        v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_c, d_s, 64, n_lead=n_lead)
    else:
        v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead)
torch.cuda.synchronize()
print("OK", octaves, nf, n_lead, n, float(d_db.abs().max()), flush=True)
