"""fp32 vs split-bf16 fused kernels (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import pitchvis_amd as P
geo = sys.argv[1] if len(sys.argv) > 1 else "48k"
pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36)) if geo == "48k" else P.VqtParameters()
v = P.Vqt(pp, 0); v.set_algo(2)
hop, nf = 256, 300
for n_lead in (0, 33000, 33001):
    pcm = (torch.rand(n_lead + hop*nf, device="cuda") - 0.5)
    out = []
    for prec in (0, 1, 2):
        v.set_algo(1 if prec == 2 else 2); v.set_gemm_precision(prec & 1)
        d_db = torch.empty((nf, v.n_bins), device="cuda"); d_cx = torch.zeros((nf, v.n_bins, 2), device="cuda")
        v.calculate_batch_db_device(pcm, hop, nf, d_db, n_lead=n_lead, d_out_cplx=d_cx)
        torch.cuda.synchronize(); out.append(d_cx.cpu().numpy().view(np.complex64)[..., 0])
    a, b, ref = out
    print('   fp32 fused vs fft path', (np.abs(a - ref) / np.abs(ref).max(axis=1, keepdims=True)).max(), ' bf16 vs fft', (np.abs(b - ref) / np.abs(ref).max(axis=1, keepdims=True)).max())
    err = np.abs(a - b) / np.abs(a).max(axis=1, keepdims=True)
    print(geo, "n_lead", n_lead, "max rel err", err.max(), "per-frame worst frames", np.argsort(err.max(axis=1))[-4:], "per-bin worst", np.argsort(err.max(axis=0))[-6:])
    e2 = np.abs(a - ref) / np.abs(ref).max(axis=1, keepdims=True)
    print('   vs fft by bin block of 12:', [float(f'{e2[:, i:i+12].max():.1e}') for i in range(0, v.n_bins, 12)])
    print("   err by bin block of 36:", [float(f"{err[:, i:i+36].max():.2e}") for i in range(0, v.n_bins, 36)])
    if n_lead == 0:
        e = np.abs(a - b)
        for f in (0, 5, 14, 15, 16, 40, 70, 100):
            k = int(np.argmax(e[f]))
            print("   frame", f, "max|ref|", np.abs(ref[f]).max(), "worst bin", k, "fp32", a[f, k], "bf16", b[f, k], "fft", ref[f, k])
    if n_lead == 33000:
        for k in (0, 1, 2, 3, 16, 17, 100, 200):
            print("   frame 50 bin", k, "blockdft", a[50, k], "fft", ref[50, k], "ratio", a[50, k] / ref[50, k])
