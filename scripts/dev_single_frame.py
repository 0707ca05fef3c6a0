"""Single-frame calls for rocprofv3 --kernel-trace --stats (developer tool): 300 x pvq_vqt_calculate_instant_db and 300 x a device-pointer
call of one frame, 48 kHz / 252 bins — the group-split FFT path (vqt_fft_frames with a workgroup per window group + db_rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pitchvis_amd as P
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
v = P.Vqt(pp, 0)
x = (np.random.default_rng(1).random(pp.n_fft, dtype=np.float32) - 0.5).astype(np.float32)
for _ in range(300): v.calculate_vqt_instant_in_db(x)
d = torch.from_numpy(x).cuda(); d_db = torch.empty((1, v.n_bins), device="cuda")
for _ in range(300): v.calculate_batch_db_device(d, pp.n_fft, 1, d_db)
torch.cuda.synchronize()
print("done")
