"""Phase stamps of the kernel-product kernel (developer tool): PVQ_STAMPS_DOTS dump -> per-phase percentiles."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "stamps_dots.bin")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["PVQ_STAMPS_DOTS"] = out
import numpy as np, torch
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0); v.set_algo(2)
if len(sys.argv) > 1: v.set_gemm_precision(int(sys.argv[1]))
hop, nf = 256, 65536
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
v.calculate_batch_db_device(d_pcm, hop, nf, d_db); torch.cuda.synchronize()
s = np.fromfile(out, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min(); tick = 10e-3
print(f"workgroups {len(s)}  span {(s[:,3].max()-t0)*tick:.1f} us")
for name, a in (("blocks (wave 0)", s[:, 1]-s[:, 0]), ("wait for all waves", s[:, 2]-s[:, 1]), ("dB phase + store", s[:, 3]-s[:, 2]), ("life", s[:, 3]-s[:, 0])):
    a = a*tick
    print(f"{name:20s} p10 {np.percentile(a,10):7.2f}  p50 {np.percentile(a,50):7.2f}  p90 {np.percentile(a,90):7.2f}  mean {a.mean():7.2f} us")
st = (s[:, 0]-t0)*tick
print("start times: p25 %.1f p50 %.1f p75 %.1f max %.1f us" % tuple(np.percentile(st, [25, 50, 75, 100])))
