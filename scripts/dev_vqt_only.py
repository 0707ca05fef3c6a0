import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
hop, nf = 256, 8192
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db)
torch.cuda.synchronize()
