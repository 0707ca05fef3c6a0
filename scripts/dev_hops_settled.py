import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import pitchvis_amd as P
from helpers import get_geom
for name, hops in (("bench_48k_252", (1600, 800, 256)), ("default_22k_588", (1600, 1344)), ("hires_96k_360", (3200,))):
    pp, _ = get_geom(name)
    for hop in hops:
        for algo in (P.ALGO_BLOCKDFT, P.ALGO_FFT):
            v = P.Vqt.new(pp, 0); v.set_algo(algo)
            nf = 65536 if algo == P.ALGO_BLOCKDFT else 16384
            d_pcm = (torch.rand(hop * nf + v.window_union, device="cuda") - 0.5) * 0.5
            d_db = torch.empty((nf, v.n_bins), device="cuda")
            words = (v.n_bins + 31) // 32
            m = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); c = torch.zeros(nf, dtype=torch.int32, device="cuda")
            ce = torch.zeros((nf, 64), device="cuda"); sz = torch.zeros((nf, 64), device="cuda")
            def step(): v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, m, c, ce, sz, 64, n_lead=v.window_union)
            for _ in range(3): step()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                step(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10): step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 10
            print(f"{name:16s} hop {hop:5d} {'block-DFT' if algo == P.ALGO_BLOCKDFT else 'FFT path '} {nf:6d} frames per call: {dt*1e3:7.3f} ms = {nf/dt/1e6:6.1f} M frames/s", flush=True)
            del v, d_pcm, d_db
