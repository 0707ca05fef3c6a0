"""Developer tool (GPU box): the peak kernels alone on white-noise dB frames of the bench geometry (or another one).
usage: python scripts/dev_peaks.py [geom] [frames] [reps] [mask|full]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pitchvis_amd as P
from helpers import get_geom

name = sys.argv[1] if len(sys.argv) > 1 else "bench_48k_252"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
mode = sys.argv[4] if len(sys.argv) > 4 else "full"
pp, op = get_geom(name)
v = P.Vqt.new(pp, 0)
hop = 128 if op.sr > 90000 else 256
g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0001)
d_pcm = (torch.rand(hop * nf, device="cuda", generator=g) - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
words = (v.n_bins + 31) // 32
d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
d_c = torch.zeros((nf, 64), device="cuda"); d_s = torch.zeros((nf, 64), device="cuda")
fn = (lambda: v.analyze_batch_device(d_db, nf, d_mask, d_cnt)) if mode == "mask" else (lambda: v.analyze_batch_device(d_db, nf, d_mask, d_cnt, d_c, d_s, 64))
fn(); torch.cuda.synchronize()
v.set_profiling(True)
t = time.perf_counter()
for _ in range(reps): fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"{name} peaks {mode}: {dt * 1e6:.1f} us per {nf} frames; kernels {v.last_kernel_ms()}; mean peaks per frame {float(d_cnt.float().mean()):.1f}")
