"""Developer tool (GPU box): frame rate of the path pitchvis_train takes (pitchvis_train/src/train.rs:30-43,128-129,341):
22 050 Hz, 7 x 36 bins, Q 10, gamma 53, one frame per three analysis-delay chunks (hop 5 952: not a power of two), PCM
resident on the device.  usage: python scripts/dev_train_rate.py [frames] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
q = 10.0
pp = P.VqtParameters(sr=22050.0, n_fft=32768, range=P.VqtRange(55.0, 7, 36), sparsity_quantile=0.999, quality=q, gamma=5.3 * q)
v = P.Vqt.new(pp, 0)
hop = 3 * P.train_chunk_samples(v)
g = torch.Generator(device="cuda"); g.manual_seed(1)
d_pcm = (torch.rand(hop * nf, device="cuda", generator=g) - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(2):
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
v.set_profiling(True)
t0 = time.perf_counter()
for _ in range(reps):
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"train geometry: hop {hop}, {nf} frames: {dt * 1e3:.3f} ms per batch = {nf / dt / 1e6:.2f} M frames/s; algo {v.last_algo()}; kernels {v.last_kernel_ms()}")
