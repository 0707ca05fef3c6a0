"""FFT path: what its parts cost (developer tool; PVQ_FFT_SKIP of the developer library: 1 skips the row dots, 2 the FFT passes, 4 runs
the unpruned passes).  usage: dev_fft_parts.py [geometry] [hop]"""
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
nf = 8192
d_pcm = (torch.rand(hop * nf + v.window_union, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=v.window_union)
torch.cuda.synchronize()
print((time.perf_counter() - t) / 10 * 1e6)
''' % (ROOT, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "bench_48k_252"
hop = sys.argv[2] if len(sys.argv) > 2 else "1600"
for skip, what in (("0", "whole kernel"), ("1", "without the row dots"), ("2", "without the FFT passes"), ("3", "gather + real split + dB only"), ("4", "unpruned passes")):
    env = dict(os.environ, PVQ_DEV_LIB="1", PVQ_FFT_SKIP=skip)
    out = subprocess.run([sys.executable, "-c", CHILD, name, hop], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        print(out.stderr[-1500:]); sys.exit(1)
    print(f"{name} hop {hop} 8192 frames, {what:32s}: {float(out.stdout.split()[-1]):8.1f} us", flush=True)
