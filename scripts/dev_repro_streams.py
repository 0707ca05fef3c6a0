"""repro of test_streams_equal_single_stream_calls_bit_for_bit[default_22k_588-1344] (round 5): which side is wrong on the second call"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import pitchvis_amd as P
from helpers import get_geom
from test_streams_gpu import _streams, _alloc, _single
name, hop = "default_22k_588", 1344
pp, _ = get_geom(name)
frames = [700, 64, 1, 300, 0, 257, 130, 999]
def run_batch(v, pcms, leads):
    o = _alloc(len(pcms), 1024, v.n_bins, 48)
    v.batch_streams_device(pcms, hop, frames, o["db"], 1024, n_leads=leads, d_peak_mask=o["mask"], d_peak_count=o["cnt"], d_center=o["ctr"], d_size=o["sz"], max_peaks=48)
    torch.cuda.synchronize()
    v.input_status()
    return o
v = P.Vqt.new(pp, 0)
leads = [0, 5000, 0, v.window_union - hop, 0, 123, 40000, 0]
pcms = _streams(len(frames), hop, frames, leads, 1000)
v.set_algo(P.ALGO_BLOCKDFT)
fresh = None
def check(rnd):
    global fresh
    o = run_batch(v, pcms, leads)
    v.set_algo(v.last_algo())
    for s in range(len(frames)):
        nf = frames[s]
        if not nf: continue
        w = _single(v, pcms[s], hop, nf, leads[s], v.n_bins, 48)
        a = int((o["db"][s, :nf] != w["db"]).sum())
        if a:
            if fresh is None:
                fresh = P.Vqt.new(pp, 0); fresh.set_algo(P.ALGO_BLOCKDFT)
            r = _single(fresh, pcms[s], hop, nf, leads[s], v.n_bins, 48)
            b = int((o["db"][s, :nf] != r["db"]).sum()); c = int((w["db"] != r["db"]).sum())
            d = (o["db"][s, :nf] != r["db"]) if b else (w["db"] != r["db"])
            rows = torch.nonzero(d.any(dim=1)).flatten().tolist(); cols = torch.nonzero(d.any(dim=0)).flatten().tolist()
            bad = o["db"][s, :nf] if b else w["db"]
            print(f"round {rnd} stream {s} ({nf} frames): batch vs single {a}; batch vs fresh single {b}; single vs fresh single {c}; bad rows {rows[:6]}..{rows[-3:]} ({len(rows)}), bad bins {cols[:6]}..{cols[-3:]} ({len(cols)}); "
                  f"sample bad {bad[rows[0], cols[0]].item()} ref {r['db'][rows[0], cols[0]].item()} nan {int(torch.isnan(bad).sum())}", flush=True)
for rnd in range(3):
    check(rnd)
    print("round", rnd, "last_algo", v.last_algo(), flush=True)
    v.set_algo(P.ALGO_AUTO)
    print("  AUTO would take", v.resolve_algo(hop, sum(frames)), "for", sum(frames), "frames; for 700:", v.resolve_algo(hop, 700), flush=True)
print("done")
