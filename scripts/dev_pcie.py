"""PCIe-inclusive rate of the host-buffer entry point (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
if not os.environ.get("PVQ_SKIP_BUILD"): g.build()
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf = 256, 65536
pcm = ((np.random.default_rng(1).random(hop * nf, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
for _ in range(2):
    out = v.calculate_batch_db(pcm, hop, nf)
t = time.perf_counter(); n = 5
for _ in range(n):
    out = v.calculate_batch_db(pcm, hop, nf)
dt = (time.perf_counter() - t) / n
print(f"host buffers (pageable numpy, 64 MiB in / 63 MiB out): {dt*1e3:.2f} ms per {nf} frames = {nf/dt/1e6:.1f} M frames/s")

pin = P.PinnedArray((hop * nf,)); pin.array[:] = pcm
pout = P.PinnedArray((nf, v.n_bins))
import ctypes as C
L = P._lib.load(); fp = C.POINTER(C.c_float)
def run_pinned():
    st = L.pvq_vqt_calculate_batch_db(v._h, pin.array.ctypes.data_as(fp), 0, hop, nf, pout.array.ctypes.data_as(fp))
    assert st == 0
for _ in range(2): run_pinned()
t = time.perf_counter()
for _ in range(n): run_pinned()
dt = (time.perf_counter() - t) / n
assert np.array_equal(pout.array, out)
print(f"host buffers (pinned via pvq_host_alloc): {dt*1e3:.2f} ms per {nf} frames = {nf/dt/1e6:.1f} M frames/s")
