import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop = 256
for n, nf in ((64, 2048), (1, 131072)):
    pcms = [stream_slice(10 + s, 0, hop * nf, "cuda") for s in range(n)]
    words = 8
    db = torch.empty((n, nf, v.n_bins), device="cuda"); m = torch.zeros((n, nf, words), dtype=torch.int32, device="cuda"); c = torch.zeros((n, nf), dtype=torch.int32, device="cuda")
    ce = torch.zeros((n, nf, 64), device="cuda"); sz = torch.zeros((n, nf, 64), device="cuda")
    def call(): v.batch_streams_device(pcms, hop, [nf] * n, db, nf, d_peak_mask=m, d_peak_count=c, d_center=ce, d_size=sz, max_peaks=64)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        call(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    # host time of a call alone
    t = time.perf_counter(); call(); th = time.perf_counter() - t; torch.cuda.synchronize()
    v.set_profiling(True)
    for _ in range(5): call()
    torch.cuda.synchronize()
    km = v.last_kernel_ms(); kn = v.last_kernel_launches(); v.set_profiling(False)
    print(f"{n} x {nf}: {dt*1e3:.3f} ms per call ({n*nf/dt/1e6:.1f} M frames/s), host side of a call {th*1e6:.0f} us; kernels: " + ", ".join(f"{k} {km[k]*1e3:.0f}us x{kn.get(k,1)//5 if kn.get(k,1)>=5 else kn.get(k,1)}" for k in km), flush=True)
