import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
from helpers import *
import torch
def run_gpu(v, pcm, hop, nf, n_lead=0):
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm, np.float32)).cuda()
    d_db = torch.full((nf, v.n_bins), -1.0, device='cuda')
    d_cx = torch.zeros((nf, v.n_bins, 2), device='cuda')
    v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead, d_out_cplx=d_cx)
    torch.cuda.synchronize()
    return d_db.cpu().numpy(), d_cx.cpu().numpy().view(np.complex64)[..., 0]
pp, op = get_geom("bench_48k_252")
v = P.Vqt.new(pp, 0)
hop = 256
for nf, n_lead in ((1, 0), (2, 16129), (63, 3), (64, 40001), (65, 0), (192, 2), (193, 16128), (194, 7), (257, 33333), (450, 1)):
    pcm = white_noise(n_lead + hop * nf, 1000 + nf)
    v.set_algo(P.ALGO_BLOCKDFT)
    db, cx = run_gpu(v, pcm, hop, nf, n_lead)
    v.set_algo(P.ALGO_FFT)
    db_f, cx_f = run_gpu(v, pcm, hop, nf, n_lead)
    e = np.abs(cx - cx_f) / np.abs(cx_f).max()
    fr, bn = np.unravel_index(e.argmax(), e.shape)
    bad = np.argwhere(e > 1e-5)
    print(nf, n_lead, "max rel err %.3g at frame %d bin %d; bad frames %s bins %s" % (e.max(), fr, bn, sorted(set(bad[:,0]))[:10], sorted(set(bad[:,1]))[:10]), flush=True)
