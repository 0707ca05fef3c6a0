#!/usr/bin/env python3
"""Developer tool (GPU box): rate of pvq_analysis_batch_preprocess_device over the number of streams (one wave owns a stream: the kernel
is latency-bound per stream and scales with streams, not frames), 252 and 588 bins, default smoothing.
usage: python3 scripts/dev_analysis_batch.py [out-file] [once [bpo]]   (once: a single 4096-stream x 128-frame call at 7 x bpo bins, for rocprofv3)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pitchvis_amd as P


def frames(n_streams, n_frames, nb, seed):
    """dB-like frames made on the device: a noise floor and a few held notes per stream"""
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.rand((n_streams, n_frames, nb), device="cuda", generator=g) * 6.0
    for j in range(5):
        b = torch.randint(3, nb - 3, (n_streams,), device="cuda", generator=g)
        lvl = 18.0 + 30.0 * torch.rand((n_streams,), device="cuda", generator=g)
        t0 = (j * n_frames) // 7
        idx = torch.arange(n_streams, device="cuda")
        x[idx, t0:, b] = lvl[:, None]
        x[idx, t0:, b - 1] = torch.maximum(x[idx, t0:, b - 1], (lvl - 9.0)[:, None])
        x[idx, t0:, b + 1] = torch.maximum(x[idx, t0:, b + 1], (lvl - 11.0)[:, None])
    return x.contiguous()


def rate(bpo, octaves, n_streams, n_frames, reps=3, outputs="all"):
    rng_ = P.VqtRange(55.0, octaves, bpo)
    nb = octaves * bpo
    x = frames(n_streams, n_frames, nb, 7)
    b = P.AnalysisBatch(rng_, n_streams)
    outs = {}
    if outputs == "all":
        words = (nb + 31) // 32
        outs = {k: torch.empty((n_streams, n_frames, nb), device="cuda") for k in ("x_vqt_smoothed", "x_vqt_peakfiltered", "x_vqt_afterglow", "calmness", "pitch_accuracy", "pitch_deviation")}
        outs["peak_mask"] = torch.zeros((n_streams, n_frames, words), dtype=torch.int32, device="cuda")
        outs["peak_count"] = torch.zeros((n_streams, n_frames), dtype=torch.int32, device="cuda")
        outs["center"] = torch.zeros((n_streams, n_frames, 64), device="cuda"); outs["size"] = torch.zeros((n_streams, n_frames, 64), device="cuda")
        outs["scene_calmness"] = torch.zeros((n_streams, n_frames), device="cuda"); outs["tuning_grid_inaccuracy"] = torch.zeros((n_streams, n_frames), device="cuda")
    elif outputs == "lean":   # what a peak / note consumer reads
        outs["peak_count"] = torch.zeros((n_streams, n_frames), dtype=torch.int32, device="cuda")
        outs["center"] = torch.zeros((n_streams, n_frames, 64), device="cuda"); outs["size"] = torch.zeros((n_streams, n_frames, 64), device="cuda")
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.preprocess_device(x, n_frames, 0.016, outs, max_peaks=64 if "center" in outs else 0)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return n_streams * n_frames / (best * 1e-3), best


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else None
    if len(sys.argv) > 2 and sys.argv[2] == "once":   # once [bpo]: 36 -> 252 bins, 84 -> 588 bins (the reference's default geometry: NK = 10, distance rule)
        print(rate(int(sys.argv[3]) if len(sys.argv) > 3 else 36, 7, 4096, 128, reps=2))
        sys.exit(0)
    lines = []
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
    for bpo, octaves in ((36, 7), (84, 7)):
        for n_streams in ((256, 4096) if quick else (64, 256, 1024, 4096, 16384)):
            n_frames = max(32, min(1000, (1 << 20) // n_streams))
            if n_streams * n_frames * bpo * octaves * 4 * 7 > 40e9:
                n_frames //= 2
            for mode in (("all",) if quick else ("all", "lean")):
                r, ms = rate(bpo, octaves, n_streams, n_frames, outputs=mode)
                lines.append(f"analysis_batch_preprocess {bpo * octaves:4d} bins, {n_streams:6d} streams x {n_frames:4d} frames, outputs {mode:4s}: {r / 1e6:8.2f} M frames/s ({ms:.2f} ms)")
                print(lines[-1], flush=True)
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        open(out, "w").write("\n".join(lines) + "\n")
