"""many streams on the FFT path: ct kernels vs the walk, dB rows (developer tool, GPU box)"""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch
import pitchvis_amd as P
from helpers import get_geom
from test_streams_gpu import _streams
pp, _ = get_geom("default_22k_588")
hop = 1344
v = P.Vqt.new(pp, 0); v.set_algo(P.ALGO_FFT)
frames = eval(os.environ.get("FRAMES", "[700, 64, 1, 300, 0, 257, 130, 999]"))
leads = eval(os.environ.get("LEADS", "[0, 5000, 0, v.window_union - hop, 0, 123, 40000, 0]"))[:len(frames)]
pcms = _streams(len(frames), hop, frames, leads, 1000)
db = torch.full((len(frames), 1024, v.n_bins), -1.0, device="cuda")
v.batch_streams_device(pcms, hop, frames, db, 1024, n_leads=leads)
torch.cuda.synchronize()
one = torch.empty((frames[0], v.n_bins), device="cuda")
v.calculate_batch_db_device(pcms[0], hop, frames[0], one, n_lead=leads[0])
torch.cuda.synchronize()
np.save(sys.argv[1], db.cpu().numpy()); np.save(sys.argv[1] + ".one.npy", one.cpu().numpy())
''' % (ROOT, ROOT)
import numpy as np
out = {}
for ct in ("0", "1"):
    f = f"/tmp/cts{ct}.npy"
    r = subprocess.run([sys.executable, "-c", CHILD, f], env=dict(os.environ, PVQ_DEV_LIB="1", PVQ_FFT_CT=ct), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-1500:]); sys.exit(1)
    out[ct] = (np.load(f), np.load(f + ".one.npy"))
(a, a1), (b, b1) = out["0"], out["1"]
print("streams call, walk vs ct: values that differ per stream:", [(s, int((a[s] != b[s]).sum())) for s in range(a.shape[0])])
print("single call, walk vs ct:", int((a1 != b1).sum()))
print("walk: streams call stream 0 vs single call:", int((a[0, :a1.shape[0]] != a1).sum()), " ct:", int((b[0, :b1.shape[0]] != b1).sum()))
d = a[0, :a1.shape[0]] != b[0, :a1.shape[0]]
if d.any():
    rows = np.nonzero(d.any(axis=1))[0]; cols = np.nonzero(d.any(axis=0))[0]
    print("stream 0 rows", rows[:8], "...", rows[-3:], len(rows), "bins", cols[:8], "...", cols[-3:], len(cols))
    r0 = rows[0]; print("row", r0, "walk", a[0, r0, cols[:6]], "ct", b[0, r0, cols[:6]], "frame max walk/ct", a[0, r0].max(), b[0, r0].max(), "min", a[0, r0].min(), b[0, r0].min())
