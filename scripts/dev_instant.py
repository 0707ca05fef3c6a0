"""Latency of the reference-shaped single-frame call, host buffer in / host vector out (developer tool):
pvq_vqt_calculate_instant_db = Vqt::calculate_vqt_instant_in_db (vqt.rs:866), the viewer's call once per rendered frame."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pitchvis_amd as P
from helpers import get_geom
for name in ("default_22k_588", "bench_48k_252", "hires_96k_360"):
    pp, _ = get_geom(name)
    v = P.Vqt.new(pp, 0)
    x = (np.random.default_rng(1).random(pp.n_fft, dtype=np.float32) - 0.5).astype(np.float32)
    for _ in range(20): v.calculate_vqt_instant_in_db(x)
    lat = []
    for _ in range(300):
        t = time.perf_counter()
        v.calculate_vqt_instant_in_db(x)
        lat.append(time.perf_counter() - t)
    lat = np.array(lat) * 1e6
    print(f"{name:16s} calculate_vqt_instant_in_db (n_fft {pp.n_fft} samples in, {v.n_bins} dB values out): median {np.median(lat):.1f} us, p10 {np.percentile(lat, 10):.1f}, p90 {np.percentile(lat, 90):.1f}", flush=True)
