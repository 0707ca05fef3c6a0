"""ct kernels vs the walk on the same frames: which bins differ (developer tool, GPU box; PVQ_DEV_LIB=1)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import pitchvis_amd as P
from helpers import get_geom, white_noise
name, hop, nf = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pp, _ = get_geom(name)
v = P.Vqt.new(pp, 0); v.set_algo(P.ALGO_FFT)
lead = int(os.environ.get("LEAD", "70000"))
pcm = torch.from_numpy(white_noise(lead + hop * nf, 5)).cuda()
cx = torch.zeros((nf, v.n_bins, 2), device="cuda"); db = torch.empty((nf, v.n_bins), device="cuda")
v.calculate_batch_db_device(pcm, hop, nf, db, n_lead=lead, d_out_cplx=cx)
torch.cuda.synchronize()
np.save(sys.argv[4], cx.cpu().numpy())
''' % (ROOT, ROOT)
name, hop, nf = sys.argv[1], sys.argv[2], sys.argv[3]
import numpy as np
out = {}
for ct in ("0", "1"):
    f = f"/tmp/ct{ct}.npy"
    r = subprocess.run([sys.executable, "-c", CHILD, name, hop, nf, f], env=dict(os.environ, PVQ_DEV_LIB="1", PVQ_FFT_CT=ct), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-1500:]); sys.exit(1)
    out[ct] = np.load(f)
a, b = out["0"], out["1"]
d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
print(f"{name} hop {hop} {nf} frames: {int(d.sum())} of {d.size} coefficients differ; frames with a difference {int(d.any(axis=1).sum())}")
bins = np.nonzero(d.any(axis=0))[0]
if bins.size:
    runs, s = [], bins[0]
    for x, y in zip(bins[:-1], bins[1:]):
        if y != x + 1: runs.append((s, x)); s = y
    runs.append((s, bins[-1]))
    print("bins that differ:", runs[:20])
    rel = np.abs(a - b).max() / np.abs(a).max()
    print("largest |difference| / largest |coefficient|:", rel)
    per_frame = d.sum(axis=1)
    print("differing bins per frame (first 16 frames):", per_frame[:16].tolist())
