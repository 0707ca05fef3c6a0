import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice
pp = P.VqtParameters(sr=22050.0, range=P.VqtRange(55.0, 5, 36), quality=1.8, gamma=4.8*1.8)
v = P.Vqt.new(pp, 0)
hop, ns, nf = 735, 1024, 128
pcms = [stream_slice(s, 0, nf*hop, "cuda") for s in range(ns)]
db = torch.empty((ns, nf, v.n_bins), device="cuda")
def streams(): v.batch_streams_device(pcms, hop, [nf]*ns, db, nf)
def loop():
    for s in range(ns): v.calculate_batch_db_device(pcms[s], hop, nf, db[s])
for name, fn in (("one streams call", streams), ("loop of calls", loop)):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(f"hop 735, 22 050 Hz / 180 bins (pitchvis_serial), {ns} streams x {nf} frames, FFT path, {name}: {ns*nf/dt/1e6:.2f} M frames/s")
