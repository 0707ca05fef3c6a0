"""A/B of a developer-library knob on the bench workload (developer tool): alternating child processes, kernel times per value.
usage: dev_ab_knob.py KNOB v1,v2,... [rounds] [iters]"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf, n = 256, 65536, int(sys.argv[1])
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
t_end = time.perf_counter() + float(os.environ.get("SETTLE_MS", "300")) * 1e-3   # until the device's clock governor has settled (profiles/r04_bench_warmup.txt)
while time.perf_counter() < t_end:
    for _ in range(8): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
    torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
v.set_profiling(True)
for _ in range(n): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
km = v.last_kernel_ms()
print(dt * 1e3, km.get("blockdft_gemm", 0.0) * 1e3, km.get("blockdft_dots_db", 0.0) * 1e3)
''' % ROOT
knob, vals = sys.argv[1], sys.argv[2].split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
iters = sys.argv[4] if len(sys.argv) > 4 else "30"
res = {x: [] for x in vals}
for r in range(rounds):
    for x in vals:
        env = dict(os.environ, PVQ_DEV_LIB="1")
        env[knob] = x
        out = subprocess.run([sys.executable, "-c", CHILD, iters], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(out.stderr[-2000:]); sys.exit(1)
        res[x].append([float(t) for t in out.stdout.split()[-3:]])
for x in vals:
    a = res[x]
    print(f"{knob}={x}: step ms " + " ".join(f"{r[0]:.4f}" for r in a) + f" (median {statistics.median(r[0] for r in a):.4f})  gemm us " +
          " ".join(f"{r[1]:.1f}" for r in a) + f" (median {statistics.median(r[1] for r in a):.1f})  dots us median {statistics.median(r[2] for r in a):.1f}", flush=True)
