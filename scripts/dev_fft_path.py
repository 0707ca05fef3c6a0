#!/usr/bin/env python3
"""Developer tool (GPU box): the FFT path's BATCH form — one kernel instantiation per window size (vqt_fft_group, round 5) against the
walk (vqt_fft_frames, PVQ_FFT_CT=0 in the developer library), alternating child processes at the settled clock.
usage: python3 scripts/dev_fft_path.py [out-file]        |  python3 scripts/dev_fft_path.py - once <geometry index>   (one timed batch, for rocprofv3)"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEOMS = [("48 kHz / 252 bins / hop 256", 48000.0, 55.0, 7, 36, 256, 16384), ("96 kHz / 360 bins / hop 128", 96000.0, 27.5, 10, 36, 128, 8192),
         ("22 050 Hz / 588 bins / hop 735", 22050.0, 55.0, 7, 84, 735, 16384), ("22 050 Hz / 180 bins / hop 735", 22050.0, 55.0, 5, 36, 735, 16384),
         ("48 kHz / 288 bins / hop 1000", 48000.0, 55.0, 8, 36, 1000, 16384)]
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
import pitchvis_amd as P
name, sr, fmin, octs, bpo, hop, nf = eval(sys.argv[1])
v = P.Vqt(P.VqtParameters(sr=sr, range=P.VqtRange(fmin, octs, bpo)), 0)
v.set_algo(P.ALGO_FFT)
d_pcm = (torch.rand(hop * nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
if len(sys.argv) > 2:   # once: for the profiler
    sys.exit(0)
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db); torch.cuda.synchronize()
n = 20
t = time.perf_counter()
for _ in range(n): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
print((time.perf_counter() - t) / n * 1e3)
''' % ROOT
if len(sys.argv) > 2 and sys.argv[2] == "once":
    subprocess.run([sys.executable, "-c", CHILD, repr(GEOMS[int(sys.argv[3])]), "once"], env=dict(os.environ), check=True)
    sys.exit(0)
lines = []
for gm in GEOMS:
    res = {"0": [], "1": []}
    for _ in range(3):
        for ct in ("0", "1"):
            out = subprocess.run([sys.executable, "-c", CHILD, repr(gm)], env=dict(os.environ, PVQ_DEV_LIB="1", PVQ_FFT_CT=ct), capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:]); sys.exit(1)
            res[ct].append(float(out.stdout.split()[-1]))
    a, b = statistics.median(res["0"]), statistics.median(res["1"])
    lines.append(f"FFT path, {gm[0]:32s} {gm[6]:6d} frames: walk (vqt_fft_frames) {a:7.3f} ms = {gm[6] / a / 1e3:6.2f} M frames/s;  per-window kernels (vqt_fft_group) {b:7.3f} ms = {gm[6] / b / 1e3:6.2f} M frames/s  ({a / b:.2f} x)")
    print(lines[-1], flush=True)
if len(sys.argv) > 1 and sys.argv[1] != "-":
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
