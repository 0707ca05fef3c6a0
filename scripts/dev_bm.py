"""128- against 256-row tiles of the fused GEMM kernels by batch size (developer tool; PVQ_FUSED_BM of the developer library).
usage: dev_bm.py hop [geometry]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
import pitchvis_amd as P
from helpers import get_geom
hop, name = int(sys.argv[1]), sys.argv[2]
pp, _ = get_geom(name)
v = P.Vqt.new(pp, 0); v.set_algo(P.ALGO_BLOCKDFT)
sizes = (64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768)
d_pcm = (torch.rand(hop * max(sizes) + v.window_union, device="cuda") - 0.5) * 0.5
d_db = torch.empty((max(sizes), v.n_bins), device="cuda")
out = []
for nf in sizes:
    for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
    torch.cuda.synchronize()
    n = 40 if nf <= 8192 else 10
    t = time.perf_counter()
    for _ in range(n): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / n * 1e6)
print(" ".join("%%.1f" %% x for x in out), float(d_db[:64].sum()))
''' % (ROOT, ROOT)
hop = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "bench_48k_252"
print("frames:            64    256    512   1024   2048   4096   8192  16384  32768")
for bm in ("256", "128", "256", "128"):
    env = dict(os.environ, PVQ_DEV_LIB="1", PVQ_FUSED_BM=bm)
    out = subprocess.run([sys.executable, "-c", CHILD, hop, name], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        print(out.stderr[-1500:]); sys.exit(1)
    vals = out.stdout.split()
    print(f"{name} hop {hop} BM {bm}: " + " ".join(f"{float(x):6.0f}" for x in vals[:-1]) + f"   (checksum {vals[-1]})", flush=True)
