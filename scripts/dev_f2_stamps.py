"""Developer tool (GPU box): per-pass timing of blockdft_fused2 from the stamps of its sampled workgroups.
usage: PVQ_F2_STAMPS=/tmp/f2.bin python scripts/dev_f2_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pitchvis_amd as P
path = os.environ.setdefault("PVQ_F2_STAMPS", "/tmp/f2_stamps.bin")
v = P.Vqt(P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36)), 0)
hop, nf = 256, 65536
d_pcm = (torch.rand(hop * nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(3):
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
v.set_profiling(True)
v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
print("sclk MHz", v.last_sclk_mhz(), v.last_kernel_ms())
h = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
h = h[h[:, 9] > 0]
tot = (h[:, 9] - h[:, 11]) / 100.0          # us per unit
nct = h[:, 10]
it = (h[:, 8] - h[:, 1]) / 100.0            # us of iteration 1 (GEMM of tile 1 beside the side work on tile 0)
passes = np.diff(np.concatenate([h[:, 4:8], h[:, 3:4]], axis=1), axis=1) / 100.0
dump = (h[:, 8] - h[:, 3]) / 100.0
for n in sorted(set(nct)):
    m = nct == n
    print(f"units of {n} column tiles: {m.sum()} sampled, {np.median(tot[m]):.1f} us per unit = {np.median(tot[m]) / (n + 1):.2f} us per iteration; "
          f"iteration 1: {np.median(it[m]):.2f} us = passes {np.median(passes[m], axis=0).round(2)} + dump {np.median(dump[m]):.2f}")
