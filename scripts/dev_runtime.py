import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "torch_first":
    import torch
    print("torch cuda avail:", torch.cuda.is_available())
import pitchvis_amd as P
try:
    v = P.Vqt(P.VqtParameters.default(), 0)
    import numpy as np
    print(mode, "ok", v.calculate_vqt_instant_in_db(np.zeros(32768, np.float32)).max())
except Exception as e:
    print(mode, "FAILED:", e)
if mode == "pvq_first":
    import torch
    print("then torch cuda avail:", torch.cuda.is_available(), torch.zeros(3).cuda().sum().item())
os.system(f"grep -E 'hip|hsa' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
