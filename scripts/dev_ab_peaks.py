"""A/B of a developer-library knob on the peak kernels (developer tool): alternating child processes, peak-kernel time per 65 536 frames of
white-noise dB frames (48 kHz / 252 bins, the bench workload's frames).  usage: dev_ab_peaks.py KNOB v1,v2,... [rounds]"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf = 256, 65536
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
words = (v.n_bins + 31) // 32
m = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); c = torch.zeros(nf, dtype=torch.int32, device="cuda")
ce = torch.zeros((nf, 64), device="cuda"); sz = torch.zeros((nf, 64), device="cuda")
for _ in range(3): v.analyze_batch_device(d_db, nf, m, c, ce, sz, 64)
torch.cuda.synchronize()
t_end = time.perf_counter() + float(os.environ.get("SETTLE_MS", "300")) * 1e-3   # until the device's clock governor has settled
while time.perf_counter() < t_end:
    for _ in range(16): v.analyze_batch_device(d_db, nf, m, c, ce, sz, 64)
    torch.cuda.synchronize()
v.set_profiling(True)
for _ in range(30): v.analyze_batch_device(d_db, nf, m, c, ce, sz, 64)
torch.cuda.synchronize()
print(v.last_kernel_ms().get("peaks_frames", 0.0) * 1e3, int(c.sum()))
''' % ROOT
knob, vals = sys.argv[1], sys.argv[2].split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {x: [] for x in vals}
for r in range(rounds):
    for x in vals:
        env = dict(os.environ, PVQ_DEV_LIB="1")
        env[knob] = x
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(out.stderr[-2000:]); sys.exit(1)
        res[x].append(out.stdout.split()[-2:])
for x in vals:
    print(f"{knob}={x}: peaks us " + " ".join(f"{float(r[0]):.1f}" for r in res[x]) + f" (median {statistics.median(float(r[0]) for r in res[x]):.1f}), peaks found {res[x][0][1]}", flush=True)
