"""What the dominant kernel's time depends on besides its own code (developer tool): the input signal (the matrix pipe's power
draw follows the operands' toggle rate) and the kernels that run between two of its launches.
usage: [SETTLE_MS=300] dev_gemm_context.py [iters]"""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time
sys.path.insert(0, %r)
import torch
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice
data, call, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf = 256, 65536
ns = hop * nf
if data == "uniform":   d_pcm = (torch.rand(ns, device="cuda") - 0.5) * 0.5
elif data == "stream":  d_pcm = stream_slice(1, 0, ns, "cuda")
elif data == "zeros":   d_pcm = torch.zeros(ns, device="cuda")
elif data == "small":   d_pcm = (torch.rand(ns, device="cuda") - 0.5) * 1e-3
elif data == "sine":    d_pcm = torch.sin(torch.arange(ns, device="cuda", dtype=torch.float32) * (2 * 3.14159265 * 440.0 / 48000.0)) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
words = (v.n_bins + 31) // 32
d_mask = torch.zeros((nf, words), device="cuda", dtype=torch.int32); d_cnt = torch.zeros(nf, device="cuda", dtype=torch.int32)
d_ctr = torch.zeros((nf, 64), device="cuda"); d_sz = torch.zeros((nf, 64), device="cuda")
def step():
    if call == "db": v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
    else: v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt, d_ctr, d_sz, 64)
for _ in range(5): step()
torch.cuda.synchronize()
settle = float(os.environ.get("SETTLE_MS", "0")) * 1e-3   # load until the clock governor has settled (0: measure right after 5 warm-up steps)
t_end = time.perf_counter() + settle
while time.perf_counter() < t_end:
    for _ in range(8): step()
    torch.cuda.synchronize()
v.set_profiling(2)
t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
km = v.last_kernel_ms(); clk = v.last_sclk_mhz()
print(dt * 1e3, km.get("blockdft_gemm", 0.0) * 1e3, clk)
''' % ROOT
iters = sys.argv[1] if len(sys.argv) > 1 else "20"
for rnd in range(2):
    for data in ("uniform", "stream", "zeros", "small", "sine"):
        for call in ("db", "analyze"):
            out = subprocess.run([sys.executable, "-c", CHILD, data, call, iters], env=dict(os.environ), capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(out.stderr[-2000:]); sys.exit(1)
            ms, g, clk = out.stdout.split()[-3:]
            print(f"round {rnd} {data:8s} {call:8s}: step {float(ms):.4f} ms  gemm {float(g):.1f} us  sclk {clk} MHz", flush=True)
