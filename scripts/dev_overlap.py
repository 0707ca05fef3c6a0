"""Do two handles on two HIP streams overlap their stages (developer tool)?  The three kernels of a step lean on different units
(matrix pipe / HBM / vector ALU); with one handle they run one after the other.  Here H handles (one per worker thread of the
reference's trainer, pitchvis_train/src/train.rs:146-155) each run K steps on a stream of their own; aggregate frames/s against H = 1.
usage: dev_overlap.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hop = 256
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))


def make(nf, seed):
    v = P.Vqt(pp, 0)
    d = dict(v=v, nf=nf, pcm=stream_slice(seed, 0, hop * nf, "cuda"), db=torch.empty((nf, v.n_bins), device="cuda"),
             mask=torch.zeros((nf, 8), device="cuda", dtype=torch.int32), cnt=torch.zeros(nf, device="cuda", dtype=torch.int32),
             ctr=torch.zeros((nf, 64), device="cuda"), sz=torch.zeros((nf, 64), device="cuda"), s=torch.cuda.Stream())
    return d


def step(d):
    d["v"].vqt_analyze_batch_device(d["pcm"], hop, d["nf"], d["db"], d["mask"], d["cnt"], d["ctr"], d["sz"], 64, stream=d["s"])


for nf in (65536, 32768, 16384):
    for H in (1, 2, 3, 4):
        hs = [make(nf, 1 + i) for i in range(H)]
        for _ in range(3):
            for d in hs: step(d)
        torch.cuda.synchronize()
        t_end = time.perf_counter() + 0.3   # until the device's clock governor has settled
        while time.perf_counter() < t_end:
            for _ in range(4):
                for d in hs: step(d)
            torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter()
            for _ in range(K):
                for d in hs: step(d)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        print(f"{nf:6d} frames per call, {H} handles/streams: {best / K * 1e3:.4f} ms per round of {H} steps = {H * nf * K / best / 1e6:.1f} M frames/s", flush=True)
        del hs
        torch.cuda.empty_cache()
