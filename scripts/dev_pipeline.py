#!/usr/bin/env python3
"""Developer tool (GPU box): the reference's default pipeline for MANY streams on the device — PCM -> pvq_vqt_calculate_batch_db_streams
(Vqt::calculate_vqt_instant_in_db per hop) -> pvq_analysis_batch_preprocess_device (AnalysisState::preprocess, default calmness-adaptive
smoothing) — frames/s of both stages together, dB frames never leaving the device.  usage: python3 scripts/dev_pipeline.py [out-file]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pitchvis_amd as P
from pitchvis_amd.sharding import stream_slice


def main():
    lines = []
    for name, sr, octaves, bpo, cases in (("48 kHz / 252 bins", 48000.0, 7, 36, ((256, 4096, 512), (800, 4096, 256), (1600, 4096, 128), (256, 256, 4096))),
                                          ("22 050 Hz / 588 bins (reference default)", 22050.0, 7, 84, ((256, 4096, 256), (1344, 4096, 128)))):
        pp = P.VqtParameters(sr=sr, range=P.VqtRange(55.0, octaves, bpo))
        v = P.Vqt.new(pp, 0)
        nb = v.n_bins
        for hop, n_streams, nf in cases:
            pcms = [stream_slice(100 + s, 0, nf * hop, "cuda") for s in range(n_streams)]
            frames = [nf] * n_streams
            d_db = torch.empty((n_streams, nf, nb), device="cuda")
            b = P.AnalysisBatch(pp.range, n_streams)
            outs = {"peak_count": torch.zeros((n_streams, nf), dtype=torch.int32, device="cuda"), "center": torch.zeros((n_streams, nf, 64), device="cuda"),
                    "size": torch.zeros((n_streams, nf, 64), device="cuda"), "x_vqt_smoothed": torch.empty((n_streams, nf, nb), device="cuda"),
                    "scene_calmness": torch.zeros((n_streams, nf), device="cuda")}
            dt = hop / sr

            def step():
                v.batch_streams_device(pcms, hop, frames, d_db, nf)
                b.preprocess_device(d_db, nf, dt, outs, max_peaks=64)

            def only_vqt():
                v.batch_streams_device(pcms, hop, frames, d_db, nf)

            res = {}
            for fn_name, fn in (("pipeline", step), ("vqt", only_vqt)):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                res[fn_name] = n_streams * nf * 5 / (time.perf_counter() - t0)
            lines.append(f"{name}, hop {hop:5d}, {n_streams} streams x {nf} frames: PCM -> VQT -> AnalysisState::preprocess {res['pipeline'] / 1e6:7.2f} M frames/s "
                         f"(the transform alone {res['vqt'] / 1e6:7.2f})")
            print(lines[-1], flush=True)
            del pcms, d_db, outs, b
            torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write("# one MI355X box; dB frames [stream][frame][bin] stay on the device between the two stages; outputs: peak count, continuous peaks, smoothed frames, scene calmness\n"
                                     + "\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
