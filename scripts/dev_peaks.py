import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.build()
import oracle as O, pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36)); op = O.OracleParams(sr=48000.0, octaves=7, buckets_per_octave=36)
v = P.Vqt(pp, 0); ov = O.OracleVqt(op)
rng = np.random.default_rng(0)
pcm = ((rng.random(256*4+20000, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
want = ov.calculate_batch(pcm, 256, 4, n_lead=20000)
mask, count, center, size = v.analyze_batch(want)
for f in range(4):
    wp, wce, wsz = O.analyze_frame(want[f], 55.0, 7, 36)
    bits = np.unpackbits(mask[f].view(np.uint8), bitorder="little")[:252]
    print("frame", f, "count", count[f])
    print(" gpu ", np.nonzero(bits)[0])
    print(" orc ", wp)
    print(" gpu c", center[f,:count[f]]); print(" orc c", wce)
    print(" gpu s", size[f,:count[f]]); print(" orc s", wsz)
