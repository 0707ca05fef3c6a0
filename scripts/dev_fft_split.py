"""FFT path, few frames: a workgroup per window group (split) against one walking the groups, by frame count (developer tool;
PVQ_FFT_SPLIT_MAX of the developer library: the largest split grid).  usage: dev_fft_split.py [geometry] [hop]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
import pitchvis_amd as P
from helpers import get_geom
name, hop = sys.argv[1], int(sys.argv[2])
pp, _ = get_geom(name)
v = P.Vqt.new(pp, 0); v.set_algo(P.ALGO_FFT)
sizes = (1, 8, 32, 64, 100, 128, 200, 256, 400, 512, 1024)
d_pcm = (torch.rand(hop * max(sizes) + v.window_union, device="cuda") - 0.5) * 0.5
d_db = torch.empty((max(sizes), v.n_bins), device="cuda")
out = []
for nf in sizes:
    for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / 50 * 1e6)
print(" ".join("%%.1f" %% x for x in out))
''' % (ROOT, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "bench_48k_252"
hop = sys.argv[2] if len(sys.argv) > 2 else "800"
print("frames:                               1      8     32     64    100    128    200    256    400    512   1024")
for mx in ("0", "100000", "0", "100000"):
    env = dict(os.environ, PVQ_DEV_LIB="1", PVQ_FFT_SPLIT_MAX=mx)
    out = subprocess.run([sys.executable, "-c", CHILD, name, hop], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        print(out.stderr[-1500:]); sys.exit(1)
    print(f"{name:16s} hop {hop:>5s} {'split' if mx != '0' else 'walk '}: " + " ".join(f"{float(x):6.1f}" for x in out.stdout.split()), flush=True)
