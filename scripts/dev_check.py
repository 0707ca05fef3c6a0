"""Developer check on a GPU box: FFT path vs oracle for several geometries + rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import oracle as O, pitchvis_amd as P

def geom(sr, mf, o, b, **kw):
    return (P.VqtParameters(sr=sr, range=P.VqtRange(mf, o, b), **kw),
            O.OracleParams(sr=sr, min_freq=mf, octaves=o, buckets_per_octave=b, **kw))

geoms = {"default": geom(22050.0, 55.0, 7, 84), "48k252": geom(48000.0, 55.0, 7, 36),
         "48k288": geom(48000.0, 55.0, 8, 36), "96k360": geom(96000.0, 27.5, 10, 36),
         "96k840": geom(96000.0, 27.5, 10, 84), "serial": geom(22050.0, 55.0, 5, 36, quality=1.8, gamma=4.8*1.8)}
algo = int(sys.argv[1]) if len(sys.argv) > 1 else P.ALGO_FFT
rng = np.random.default_rng(0)
for name, (pp, op) in geoms.items():
    v = P.Vqt(pp, 0); ov = O.OracleVqt(op)
    v.set_algo(algo)
    hop = 256 if pp.sr < 90000 else 128
    nf = 200
    pcm = ((rng.random(hop*nf + 5000, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    n_lead = 5000
    d_pcm = torch.from_numpy(pcm).cuda()
    d_db = torch.empty((nf, v.n_bins), device="cuda"); d_c = torch.empty((nf, v.n_bins, 2), device="cuda")
    try:
        v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=n_lead, d_out_cplx=d_c)
    except P.PvqError as e:
        print(name, 'not applicable:', e); continue
    torch.cuda.synchronize()
    got = d_db.cpu().numpy(); gc = d_c.cpu().numpy().view(np.complex64)[..., 0]
    want, wc = ov.calculate_batch(pcm, hop, nf, n_lead=n_lead, want_complex=True)
    mag_err = np.abs(np.abs(gc) - np.abs(wc)).max() / np.abs(wc).max()
    rel = (np.abs(np.abs(gc) - np.abs(wc)) / np.abs(wc)).max()
    print(f"{name:8s} algo={v.last_algo()} dB max err {np.abs(got-want).max():.2e}  mag err/max {mag_err:.2e}  per-bin rel max {rel:.2e}")
    # single frame API
    x = O.test_create_sines(op, [440.0, 554.37])
    v.set_algo(P.ALGO_AUTO)
    a = v.calculate_vqt_instant_in_db(x); b = ov.calculate_vqt_instant_in_db(x)
    print(f"         instant: max err {np.abs(a-b).max():.2e}  argmax {a.argmax()} {b.argmax()}")
    # peaks on oracle frames
    fa = v.analyze_frames(want)
    bad = 0
    for f in range(nf):
        wp, wce, wsz = O.analyze_frame(want[f], op.min_freq, op.octaves, op.buckets_per_octave)
        gp = sorted(fa[f].peaks)
        if list(wp) != gp: bad += 1
        else:
            ce = np.array([p.center for p in fa[f].peaks_continuous]); sz = np.array([p.size for p in fa[f].peaks_continuous])
            if len(ce) and (np.abs(ce-wce).max() > 1e-3 or np.abs(sz-wsz).max() > 1e-3): bad += 1
    print(f"         peaks: {bad} bad frames of {nf}; avg peaks {np.mean([len(f.peaks) for f in fa]):.1f}")

# timing 48k252
pp, op = geoms["48k252"]
v = P.Vqt(pp, 0); v.set_algo(algo); v.set_profiling(True)
hop, nf = 256, 65536
d_pcm = (torch.rand(hop*nf, device="cuda") - 0.5) * 0.5
d_db = torch.empty((nf, v.n_bins), device="cuda")
words = (v.n_bins+31)//32
d_mask = torch.zeros((nf, words), dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(nf, dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    v.vqt_analyze_batch_device(d_pcm, hop, nf, d_db, d_mask, d_cnt)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"iter {it}: {dt*1e3:.2f} ms -> {nf/dt/1e6:.2f} M frames/s  kernels {v.last_kernel_ms()}")
