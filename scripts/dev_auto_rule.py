"""Does PVQ_ALGO_AUTO pick the faster path (developer tool)?  Per geometry / hop / batch size: time per call of the FFT path, the
block-DFT path and what AUTO resolves to; a '!' marks a choice more than 15 % slower than the other path."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pitchvis_amd as P
from helpers import get_geom

def t_call(v, d_pcm, hop, nf, d_db, lead):
    for _ in range(3): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=lead)
    torch.cuda.synchronize()
    n = 40 if nf <= 4096 else 10
    t = time.perf_counter()
    for _ in range(n): v.calculate_batch_db_device(d_pcm, hop, nf, d_db, n_lead=lead)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

bad = 0
for name, hops in (("bench_48k_252", (800, 1600, 3200, 320)), ("default_22k_588", (1600, 1344, 704)), ("serial_22k_180", (704, 1472)),
                   ("hires_96k_360", (3200, 1600)), ("bench_48k_288", (1600,))):
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    for hop in hops:
        sizes = (64, 256, 1024, 2048, 4096, 8192, 16384)
        d_pcm = (torch.rand(hop * max(sizes) + v.window_union, device="cuda") - 0.5) * 0.5
        d_db = torch.empty((max(sizes), v.n_bins), device="cuda")
        lead = v.window_union
        v.set_algo(P.ALGO_AUTO)
        first_block = next((n for n in range(64, 1 << 20, 64) if v.resolve_algo(hop, n) == P.ALGO_BLOCKDFT), None)
        line = []
        for nf in sizes:
            v.set_algo(P.ALGO_AUTO); pick = v.resolve_algo(hop, nf)
            v.set_algo(P.ALGO_FFT); tf = t_call(v, d_pcm, hop, nf, d_db, lead)
            v.set_algo(P.ALGO_BLOCKDFT); tb = t_call(v, d_pcm, hop, nf, d_db, lead)
            chosen, other = (tb, tf) if pick == P.ALGO_BLOCKDFT else (tf, tb)
            flag = "!" if chosen > 1.15 * other else " "
            bad += flag == "!"
            line.append(f"{nf}: fft {tf:.0f} blk {tb:.0f} -> {'blk' if pick == P.ALGO_BLOCKDFT else 'fft'}{flag}")
        print(f"{name:16s} hop {hop:5d} (AUTO: block-DFT from {first_block} frames)  " + " | ".join(line), flush=True)
print("choices more than 15 % off:", bad)
