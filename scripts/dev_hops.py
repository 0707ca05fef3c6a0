#!/usr/bin/env python3
"""Developer tool (GPU box): frames/s of PCM -> dB -> peaks at hops that do not divide the windows, general block-DFT form against the FFT
path, alternating on one box.  usage: python3 scripts/dev_hops.py [out-file]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice


def rate(v, algo, hop, nf, d_pcm, bufs, reps):
    v.set_algo(algo)
    def step():
        v.vqt_analyze_batch_device(d_pcm, hop, nf, *bufs, 64)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return nf * reps / (time.perf_counter() - t0)


def main():
    lines = []
    for name, sr, octaves, bpo, hops in (("48 kHz / 252 bins", 48000.0, 7, 36, (1600, 800, 3200, 1280, 256)), ("22 050 Hz / 588 bins (reference default)", 22050.0, 7, 84, (1600, 704, 256)),
                                         ("96 kHz / 360 bins", 96000.0, 10, 36, (3200,))):
        pp = P.VqtParameters(sr=sr, range=P.VqtRange(55.0 if sr < 90000 else 27.5, octaves, bpo))
        v = P.Vqt.new(pp, 0)
        nb, words = v.n_bins, (v.n_bins + 31) // 32
        for hop in hops:
            nf = 32768
            d_pcm = stream_slice(7, 0, nf * hop, "cuda")
            bufs = (torch.empty((nf, nb), device="cuda"), torch.zeros((nf, words), dtype=torch.int32, device="cuda"), torch.zeros(nf, dtype=torch.int32, device="cuda"),
                    torch.zeros((nf, 64), device="cuda"), torch.zeros((nf, 64), device="cuda"))
            rb, rf = [], []
            for _ in range(2):
                rb.append(rate(v, P.ALGO_BLOCKDFT, hop, nf, d_pcm, bufs, 5))
                rf.append(rate(v, P.ALGO_FFT, hop, nf, d_pcm, bufs, 3))
            v.set_algo(P.ALGO_BLOCKDFT); v.set_profiling(True)
            v.vqt_analyze_batch_device(d_pcm, hop, nf, *bufs, 64); torch.cuda.synchronize()
            ms = v.last_kernel_ms(); fl = v.last_gemm_flop(); v.set_profiling(False)
            lines.append(f"{name}, hop {hop:5d}, {nf} frames: block-DFT {max(rb) / 1e6:7.2f} M frames/s | FFT path {max(rf) / 1e6:7.2f} M frames/s | x {max(rb) / max(rf):.2f} | "
                         f"kernels ms {({k: round(x, 3) for k, x in ms.items()})}, GEMM {fl / (ms.get('blockdft_gemm', 1e9) * 1e-3) / 1e12:.1f} TFLOP/s executed")
            print(lines[-1], flush=True)
            del d_pcm, bufs
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("# PCM -> dB -> peaks (mask + count + continuous), fp32, best of 2 alternating rounds, one box\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
