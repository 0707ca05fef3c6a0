"""Time per call against the batch size (developer tool): where the fixed costs of a call sit.
usage: dev_small_batches.py [hop] [algo: 0 auto, 1 fft, 2 block-DFT]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
if len(sys.argv) > 2: v.set_algo(int(sys.argv[2]))
nmax = 65536
d_pcm = (torch.rand(hop * nmax + v.window_union, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nmax, v.n_bins), device="cuda")
d_mask = torch.zeros((nmax, 8), device="cuda", dtype=torch.int32); d_cnt = torch.zeros(nmax, device="cuda", dtype=torch.int32)
d_ctr = torch.zeros((nmax, 64), device="cuda"); d_sz = torch.zeros((nmax, 64), device="cuda")
for nf in ((1, 16, 64, 256, 1024, 4096, 16384, 65536) if len(sys.argv) <= 2 else (64, 128, 256, 512, 1024, 2048, 4096, 8192)):
    for what in ("db", "analyze"):
        def step():
            if what == "db": v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union - hop)
            else: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_ctr, d_sz, 64, n_lead=v.window_union - hop)
        for _ in range(5): step()
        torch.cuda.synchronize()
        n = 200 if nf <= 4096 else 30
        t = time.perf_counter()
        for _ in range(n): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        # one call at a time (latency): synchronise after each
        t = time.perf_counter()
        for _ in range(50): step(); torch.cuda.synchronize()
        lat = (time.perf_counter() - t) / 50
        v.set_profiling(True); step(); torch.cuda.synchronize(); km = v.last_kernel_ms(); kn = v.last_kernel_launches(); v.set_profiling(False)
        ks = " ".join(f"{k}={km[k]*1e3:.0f}us x{kn.get(k,1)}" for k in km)
        print(f"hop {hop} {nf:6d} frames {what:8s}: {dt*1e6:8.1f} us per call back to back ({nf/dt/1e6:7.2f} M frames/s), {lat*1e6:8.1f} us alone  algo {v.last_algo()}  [{ks}]", flush=True)
