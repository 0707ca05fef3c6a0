"""Determinism soak of round 4's paths (developer tool): general hops (blockdft_gemm_gen, 256- and 128-row tiles), interleaved grids,
the FFT path at mid-size batches, the many-streams call — the same batch N times, every output compared bit for bit."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pitchvis_amd as P
from helpers import get_geom
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = 0
CASES = [("bench_48k_252", 1600, 20000), ("bench_48k_252", 1600, 3000), ("bench_48k_252", 800, 9000), ("bench_48k_252", 320, 5000), ("bench_48k_252", 800, 900),
         ("default_22k_588", 1344, 12000), ("default_22k_588", 1344, 2000), ("hires_96k_360", 3200, 4000), ("serial_22k_180", 735, 6000), ("bench_48k_288", 256, 3000)]
for name, hop, nf in CASES:
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    gen = torch.Generator(device="cuda"); gen.manual_seed(4321)
    d_pcm = (torch.rand(hop * nf + 555, device="cuda", generator=gen) - 0.5) * 0.5
    words = (v.n_bins + 31) // 32
    # the same samples once more as 7 ragged streams in one call
    cuts = [0, nf // 9, nf // 4, nf // 4 + 1, nf // 2, nf - nf // 5, nf - 3, nf]
    frames = [cuts[i + 1] - cuts[i] for i in range(7)]
    pcms = [d_pcm[555 + hop * cuts[i]: 555 + hop * cuts[i + 1]].clone() for i in range(7)]
    ref = None
    for it in range(runs):
        o = dict(db=torch.empty((nf, v.n_bins), device="cuda"), mask=torch.zeros((nf, words), dtype=torch.int32, device="cuda"),
                 cnt=torch.zeros(nf, dtype=torch.int32, device="cuda"), c=torch.zeros((nf, 64), device="cuda"), s=torch.zeros((nf, 64), device="cuda"))
        v.vqt_analyze_batch_device(d_pcm, hop, nf, o["db"], o["mask"], o["cnt"], o["c"], o["s"], 64, n_lead=555)
        sdb = torch.empty((7, max(frames), v.n_bins), device="cuda")
        scnt = torch.zeros((7, max(frames)), dtype=torch.int32, device="cuda"); smask = torch.zeros((7, max(frames), words), dtype=torch.int32, device="cuda")
        v.batch_streams_device(pcms, hop, frames, sdb, max(frames), d_peak_mask=smask, d_peak_count=scnt)
        torch.cuda.synchronize()
        cur = tuple(o.values()) + (sdb, scnt, smask)
        if ref is None:
            ref = cur
        else:
            for k, (a, b) in enumerate(zip(ref, cur)):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"MISMATCH {name} hop {hop} nf {nf} run {it} output {k}: {int((a != b).sum())} entries", flush=True)
    print(f"{name:16s} hop {hop:5d} {nf:6d} frames (path {v.resolve_algo(hop, nf)}; streams call path {v.last_algo()}): {runs} runs " + ("ok" if bad == 0 else f"- {bad} mismatches so far"), flush=True)
print("SOAK", "CLEAN" if bad == 0 else f"FAILED {bad}")
