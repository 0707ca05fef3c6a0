#!/usr/bin/env python3
"""Developer tool (GPU box): the many-streams entry point against the single-stream one on ONE box, alternating — N streams x F frames in
ONE pvq_vqt_analyze_batch_streams call, the same streams as N single-stream calls, and one stream of N x F frames (48 kHz, 252 bins,
hop 256, PCM -> dB -> peaks with continuous output).  usage: python3 scripts/dev_streams.py [out-file]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice

HOP = 256


def bench(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.3   # until the device's clock governor has settled (profiles/r04_bench_warmup.txt)
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
    v = P.Vqt.new(pp, 0)
    nb, words, mp = v.n_bins, (v.n_bins + 31) // 32, 64
    lines = []
    for n_streams, nf in ((64, 2048), (256, 512), (1024, 128), (2, 65536), (16, 8192), (32, 4096)):
        total = n_streams * nf
        pcms = [stream_slice(0x5EED0001 + s, 0, nf * HOP, "cuda") for s in range(n_streams)]
        one = stream_slice(0x5EED0001, 0, total * HOP, "cuda")
        db = torch.empty((n_streams, nf, nb), device="cuda")
        mask = torch.zeros((n_streams, nf, words), dtype=torch.int32, device="cuda"); cnt = torch.zeros((n_streams, nf), dtype=torch.int32, device="cuda")
        ctr = torch.zeros((n_streams, nf, mp), device="cuda"); sz = torch.zeros((n_streams, nf, mp), device="cuda")
        frames = [nf] * n_streams

        def streams():
            v.batch_streams_device(pcms, HOP, frames, db, nf, d_peak_mask=mask, d_peak_count=cnt, d_center=ctr, d_size=sz, max_peaks=mp)

        def loop():
            for s in range(n_streams):
                v.vqt_analyze_batch_device(pcms[s], HOP, nf, db[s], mask[s], cnt[s], ctr[s], sz[s], mp)

        def single():
            v.vqt_analyze_batch_device(one, HOP, total, db.view(total, nb), mask.view(total, words), cnt.view(total), ctr.view(total, mp), sz.view(total, mp), mp)

        res = {}
        for rnd in range(2):   # alternating: the box's clock drifts
            for name, fn in (("single", single), ("streams", streams), ("loop", loop)):
                if name == "loop" and n_streams > 256 and rnd:
                    continue
                res.setdefault(name, []).append(bench(fn, reps=10 if name == "loop" else 20))
        r = {k: total / min(t) / 1e6 for k, t in res.items()}
        lines.append(f"{n_streams:5d} streams x {nf:6d} frames ({total} frames per step): one stream of all frames {r['single']:7.2f} M frames/s | "
                     f"ONE streams call {r['streams']:7.2f} M frames/s ({r['streams'] / r['single']:.3f} x) | {n_streams} single-stream calls {r['loop']:7.2f} M frames/s")
        print(lines[-1], flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("# 48 kHz / 252 bins / hop 256, PCM -> dB -> peaks (mask + count + continuous), fp32 MFMA, best of 2 x 20 steps per form, one box\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
