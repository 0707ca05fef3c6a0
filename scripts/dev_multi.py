#!/usr/bin/env python3
"""Developer tool (GPU box): what the multi-device driver (pvq_vqt_analyze_batch_multi: host arrays in and out) costs over the device-pointer
entry point on ONE GPU — k = 1, 2, 4 handles on device 0, pageable and page-locked (pvq_host_alloc) caller arrays, against
vqt_analyze_batch_device on resident buffers.  usage: python3 scripts/dev_multi.py [out-file]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pitchvis_amd as P
import ctypes as C


def main():
    pp = P.VqtParameters(sr=48000.0, range=P.VqtRange(55.0, 7, 36))
    hop, nf, mp = 256, 65536, 64
    handles = [P.Vqt.new(pp, 0) for _ in range(4)]
    nb, words = handles[0].n_bins, (handles[0].n_bins + 31) // 32
    rng = np.random.default_rng(1)
    pcm = ((rng.random(hop * nf, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    lines = []
    d_pcm = torch.from_numpy(pcm).cuda()
    bufs = (torch.empty((nf, nb), device="cuda"), torch.zeros((nf, words), dtype=torch.int32, device="cuda"), torch.zeros(nf, dtype=torch.int32, device="cuda"),
            torch.zeros((nf, mp), device="cuda"), torch.zeros((nf, mp), device="cuda"))
    def dev():
        handles[0].vqt_analyze_batch_device(d_pcm, hop, nf, *bufs, mp)
        torch.cuda.synchronize()
    for _ in range(3): dev()
    t0 = time.perf_counter()
    for _ in range(10): dev()
    t_dev = (time.perf_counter() - t0) / 10
    lines.append(f"device-pointer entry point, resident buffers: {t_dev * 1e3:.3f} ms per {nf} frames = {nf / t_dev / 1e6:.1f} M frames/s")
    for k in (1, 2, 4):
        for _ in range(2):
            P.Vqt.analyze_batch_multi(handles[:k], pcm, hop, nf, max_peaks=mp)
        t0 = time.perf_counter()
        for _ in range(5):
            P.Vqt.analyze_batch_multi(handles[:k], pcm, hop, nf, max_peaks=mp)
        t = (time.perf_counter() - t0) / 5
        lines.append(f"pvq_vqt_analyze_batch_multi, k = {k} handles on device 0, pageable host arrays (64 MiB in, 63 + 36 MiB out): {t * 1e3:.2f} ms = {nf / t / 1e6:.1f} M frames/s")
    # page-locked caller arrays (pvq_host_alloc): the same call on pinned input / dB output
    pin_in, pin_out = P.PinnedArray((hop * nf,)), P.PinnedArray((nf, nb))
    pin_in.array[:] = pcm
    L = handles[0]._L
    fpt = C.POINTER(C.c_float)
    ap = P.AnalysisParameters()._c()
    for k in (1, 2, 4):
        hs = (C.c_void_p * k)(*[h._h for h in handles[:k]])
        def call():
            st = L.pvq_vqt_analyze_batch_multi(hs, k, pin_in.array.ctypes.data_as(fpt), 0, hop, nf, C.byref(ap), pin_out.array.ctypes.data_as(fpt), None, None, None, None, 0)
            assert st == 0
        for _ in range(2): call()
        t0 = time.perf_counter()
        for _ in range(5): call()
        t = (time.perf_counter() - t0) / 5
        lines.append(f"pvq_vqt_analyze_batch_multi, k = {k}, page-locked arrays (pvq_host_alloc), dB only: {t * 1e3:.2f} ms = {nf / t / 1e6:.1f} M frames/s")
    print("\n".join(lines))
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("# one MI355X box, 48 kHz / 252 bins / hop 256, 65 536 frames per call\n" + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
