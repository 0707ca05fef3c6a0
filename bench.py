#!/usr/bin/env python3
"""bench.py — VQT frames/s on MI355X (BASELINE.json metric), with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--algo auto|fft|blockdft]

One "step" = one pass of the hot path (PCM -> per-window-group spectra -> sparse kernel products
-> frame-relative dB -> peak/note detection) over one batch of synthetic input resident in HBM:
BASELINE.json configs[1], "65 536-hop batch of 48 kHz mono white noise", geometry
VqtParameters{sr 48000, n_fft 32768, range{55 Hz, 7 oct, 36 bins/oct}, sparsity 0.999, Q 1.6,
gamma 7.68}, hop 256, 252 bins, fp32.

`value` is measured on THAT workload at every N (BASELINE.json quotes its metric on 48 kHz / hop 256 /
252 bins), so that the driver's 1 -> 8 curve compares like with like: N > 1 (launched by
torch.distributed.run, one rank per GPU) analyses ONE hop stream of N x 65 536 hops of the same
white noise (a counter-based function of (seed, sample index), so every rank generates exactly its
own piece); rank r takes the contiguous frame range plan_shard gives it together with its real
window-union halo (the 16 128 samples before its first hop); kernel tables are replicated; there
is no collective on the data path.  Weak scaling: every rank processes F frames,
value = N*F*K / max-over-ranks time.

BASELINE.json configs[2] ("1 M-hop batch sharded across 8 x MI355X, 8 octaves x 36 bins/oct":
288 bins, 131 072 hops per GPU, seed 0x5EED0003, real halos) is measured in the same run in a
timed region of its own (same K steps, same barriers, max over ranks) and reported under the
extra key "config2" of the same line — at N = 1 too ("one GPU's share"), so that either
workload's efficiency can be computed from per-N values of ONE workload.  `--config 2` makes
configs[2] the headline instead (developer use).

The timed region and the device's clock.  An idle MI355X needs tens of milliseconds of load before its clock governor holds the
sustained shader clock; W = 5 warm-up steps of this workload are 2.5 ms and K = 20 timed steps 10 ms, i.e. a window that lies
wholly inside the ramp (measured on one box, same build: 131 M frames/s and a 305 us dominant kernel right after 5 warm-up steps,
146-153 M and 262 us after 25 ms - 1 s of load, profiles/r04_bench_warmup.txt).  `value` is the SUSTAINED rate: the same step
runs untimed for --settle-ms (default 300 ms) first, then the W warm-up steps, then EXACTLY K timed steps between barrier +
synchronize on both sides.  The window a cold device gives — W warm-up steps from idle, then K timed steps, the first thing
the process measures — is in the same line as "cold_start"; `--settle-ms 0` makes it the headline.

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md §8(d), BASELINE.md §2 — per-frame algorithmic work: FFT-route flop (rFFTs at 2.5 N log2 N + 8 flop per kernel
# non-zero + dB/peaks) and compulsory bytes (hop*4 new input + n_bins*4 dB out + ceil(n_bins/32)*4 peak mask)
WORKLOADS = {
    # BASELINE configs[1]: the single-GPU bench line
    1: dict(octaves=7, n_bins=252, frames=65536, seed=0x5EED0001, f_alg=1.12e6, b_alg=2064.0,
            name="BASELINE configs[1]: {F}-hop batch of 48 kHz mono white noise per GPU, hop 256, "
                 "VqtParameters{{sr 48000, n_fft 32768, 55 Hz, 7 oct x 36 = 252 bins, sparsity 0.999, "
                 "Q 1.6, gamma 7.68}}; PCM -> VQT dB frames -> peaks (mask+count+continuous)"),
    # BASELINE configs[2]: the sharded multi-GPU case
    2: dict(octaves=8, n_bins=288, frames=131072, seed=0x5EED0003, f_alg=1.14e6, b_alg=2212.0,
            name="BASELINE configs[2]: one {T}-hop stream of 48 kHz mono white noise (seed 0x5EED0003) sharded over {N} GPUs, "
                 "{F} hops per GPU with their window-union halo, hop 256, VqtParameters{{sr 48000, n_fft 32768, 55 Hz, "
                 "8 oct x 36 = 288 bins, sparsity 0.999, Q 1.6, gamma 7.68}}; PCM -> VQT dB frames -> peaks (mask+count+continuous)"),
}
PEAK_FP32_TFLOPS = 157.3            # MI355X fp32 dense MFMA peak = fp32 vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak (MI355X_MICROARCH.md); the split form spends 6 bf16 products per fp32 product
PEAK_HBM_GBS = 8000.0               # HBM3E spec peak

SR, HOP = 48000.0, 256


def _cpu_worker(args):
    """one oracle instance per worker over a disjoint frame range (train.rs:146-155 pattern)"""
    seed, n_frames, lib = args
    if lib:
        os.environ["PVQ_ORACLE_LIB"] = lib   # (spawned worker: set before the oracle module loads its library)
    import numpy as np
    import oracle as O
    op = O.OracleParams(sr=SR, min_freq=55.0, octaves=7, buckets_per_octave=36)
    ov = O.OracleVqt(op)
    rng = np.random.default_rng(seed)
    n_lead = 16384
    pcm = ((rng.random(n_lead + n_frames * HOP, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    t0 = time.perf_counter()
    db = ov.calculate_batch(pcm, HOP, n_frames, n_lead=n_lead)
    n_peaks = 0
    for f in range(n_frames):
        idx, _, _ = O.analyze_frame(db[f], 55.0, 7, 36)
        n_peaks += idx.size
    return time.perf_counter() - t0, n_peaks


def _cpu_default_geometry_ms(n_frames=1500):
    """one thread, the reference's DEFAULT geometry (22 050 Hz, 7 x 84 = 588 bins), transform only: the figure VQT_REVIEW.md:363-365
    publishes for the Rust crate (0.091 ms per frame, rustfft, an unspecified desktop CPU)"""
    import numpy as np
    import oracle as O
    ov = O.OracleVqt(O.OracleParams())
    rng = np.random.default_rng(5)
    pcm = ((rng.random(16384 + n_frames * HOP, dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        ov.calculate_batch(pcm, HOP, n_frames, n_lead=16384)
        best = min(best, time.perf_counter() - t0)
    return best / n_frames * 1e3


def cpu_baseline():
    """The CPU restatement of the reference path (oracle, kind "port") on this host's cores, bounded sample: one worker per available
    core x 2048 frames of the same workload (white noise, 48 kHz / 252 bins, transform + dB + peaks), SURVEY.md 8d: one thread, all
    cores (-O2 as the reference's release profile), all cores at -O3 -march=native (courtesy figure), and the one-thread time per frame
    at the reference's default geometry next to the 0.091 ms it publishes for its rustfft path."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import subprocess
    import oracle as O
    O.build()
    orc_dir = os.path.join(ROOT, "oracle")
    native = os.path.join(orc_dir, "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-C", orc_dir, "liboracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        native = None
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # every core this process may actually run on: the affinity mask, capped by the cgroup's CPU quota where there is one (a GPU
    # box of this pool shows 256 cores in the mask and grants 16)
    quota = None
    try:
        q, per_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per_))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, int(round(q / per_)))
        except Exception:
            quota = None
    cores = max(1, min(avail, quota or avail, 256))
    per = 2048
    t1, _ = _cpu_worker((1000, per, None))   # single thread first
    ms_default = _cpu_default_geometry_ms()

    def all_cores(lib):
        with cf.ProcessPoolExecutor(max_workers=cores, mp_context=mp.get_context("spawn")) as ex:
            list(ex.map(_cpu_worker, [(1, 8, lib)] * cores))  # start the workers, load the library
            t0 = time.perf_counter()
            list(ex.map(_cpu_worker, [(2000 + i, per, lib) for i in range(cores)]))
            return cores * per / (time.perf_counter() - t0)

    v_o2 = all_cores(None)
    v_native = all_cores(native) if native and os.path.exists(native) else None
    return {
        "value": round(v_o2, 1),
        "unit": "frames/s",
        "cores": cores,
        "cores_available": avail,
        "cpu_quota": quota,
        "kind": "port",
        "sample": f"{cores} workers (every available core) x {per} frames (hop 256, 48 kHz/252 bins, white noise), VQT+dB+peaks, "
                  f"oracle/pvq_oracle.c -O2 (the reference's release profile); 1 core: {per / t1:.0f} frames/s",
        "value_1core": round(per / t1, 1),
        "value_native_O3": round(v_native, 1) if v_native else None,
        "default_geometry_ms_per_frame_1core": round(ms_default, 4),
        "reference_published_ms_per_frame": 0.091,
        "note": "the port, not the reference's binary (Rust, cannot be built here): scalar radix-2 real FFT, two stages per pass, where the "
                f"reference uses rustfft's SIMD kernels; {ms_default:.3f} ms per frame on one core of THIS host at the reference's default "
                "22 050 Hz / 588-bin geometry against the 0.091 ms VQT_REVIEW.md:363-365 publishes for the crate on an unspecified desktop "
                "CPU; value_native_O3: the same source at -O3 -march=native on the same cores; cores = min(affinity mask, cgroup CPU quota)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0, help="frames (hops) per GPU per step (default: the configuration's own — 65 536 for configs[1], 131 072 for configs[2])")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2],
                    help="BASELINE config the headline `value` is measured on: 1 (252 bins, 65 536 hops per GPU: the default at EVERY N, what "
                         "BASELINE.json's metric is quoted on) or 2 (288 bins, 131 072 hops per GPU); the other one is reported under an extra key")
    ap.add_argument("--algo", default="auto", choices=["auto", "fft", "blockdft"])
    ap.add_argument("--gemm", default="f32", choices=["f32", "bf16x3"],
                    help="arithmetic of the block-DFT GEMM / kernel product: fp32 MFMA (default, the library default) or the exact "
                         "3-way bf16 split on the bf16 matrix cores; the other one is measured too and reported as alt_gemm")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed load before the warm-up steps so that the timed region sees the device's sustained clock (0: none; the cold window is reported as cold_start either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-config", action="store_true", help="skip the second BASELINE configuration (the extra key config2 / config1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the plumbing on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    # CPU baseline first (rank 0, N=1 only), before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    # one rank at a time through build(): on a cold tree the first one compiles, the others find the stamped libraries (eight
    # concurrent `make -B` into one directory otherwise)
    import fcntl
    with open(os.path.join(ROOT, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            entry.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    import pitchvis_amd as P
    from pitchvis_amd.sharding import plan_shard, stream_slice

    device_index = local_rank % max(torch.cuda.device_count(), 1)   # identity on an N-GPU node
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm

    def measure(W, F, with_alt):
        """one workload, this rank's shard: warm-up, then EXACTLY args.steps timed steps between barrier + synchronize on both sides
        (max over ranks), then the untimed per-kernel passes"""
        n_bins = W["n_bins"]
        params = P.VqtParameters(sr=SR, range=P.VqtRange(55.0, W["octaves"], 36))
        vqt = P.Vqt.new(params, device=device_index)
        vqt.set_algo({"auto": P.ALGO_AUTO, "fft": P.ALGO_FFT, "blockdft": P.ALGO_BLOCKDFT}[args.algo])
        vqt.set_gemm_precision(P.GEMM_BF16X3 if args.gemm == "bf16x3" else P.GEMM_F32)
        assert vqt.n_bins == n_bins
        # this rank's shard of ONE world*F-frame stream: its hops plus the window-union halo that precedes them.  The stream is a
        # counter-based function of (seed, sample index), so every rank computes exactly its own piece — halo included — of one
        # signal without materialising the rest (configs[2] at N = 8: the whole stream is 1 GiB).
        shard = plan_shard(world * F, HOP, vqt.window_union, rank, world)
        assert shard.n_frames == F
        d_pcm = stream_slice(W["seed"], shard.sample_begin, shard.sample_end, "cuda")
        torch.cuda.empty_cache()
        d_db = torch.empty((F, n_bins), device="cuda", dtype=torch.float32)
        words = (n_bins + 31) // 32
        max_peaks = 64
        d_mask = torch.zeros((F, words), device="cuda", dtype=torch.int32)
        d_cnt = torch.zeros(F, device="cuda", dtype=torch.int32)
        d_ctr = torch.zeros((F, max_peaks), device="cuda", dtype=torch.float32)
        d_sz = torch.zeros((F, max_peaks), device="cuda", dtype=torch.float32)

        def step():
            vqt.vqt_analyze_batch_device(d_pcm, HOP, shard.n_frames, d_db, d_mask, d_cnt, d_ctr, d_sz, max_peaks,
                                         n_lead=shard.n_lead)

        def timed_region():
            """W untimed warm-up steps, then EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks"""
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            vqt.set_profiling(2)   # HIP events around the dominant kernel's launches only, on the launch stream (every event record costs the stream ~3 us: the other kernels are timed in an untimed pass below)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0
            km = vqt.last_kernel_ms()
            clk = vqt.last_sclk_mhz()   # shader clock inside the dominant kernel's K loop, sampled in the region's last launch
            vqt.set_profiling(False)
            per_rank = [[dt_, float(clk), float(km.get("blockdft_gemm", 0.0))]]
            if world > 1:
                dev_ = "cuda" if args.backend == "nccl" else "cpu"
                mine = torch.tensor(per_rank[0], device=dev_, dtype=torch.float64)
                every = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)   # (after the timed region: every rank's own clock around the same K steps, its sampled shader clock, its dominant kernel)
                per_rank = [[float(x) for x in e.tolist()] for e in every]
                t = torch.tensor([dt_], device=dev_, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_ = float(t.item())
            timed_region.per_rank = per_rank
            return dt_, km, clk

        # the window a cold device gives (the first thing this process measures on it) ...
        dt_cold, km_cold, clk_cold = timed_region()
        # ... then load until the clock governor has settled, and the window that counts
        if args.settle_ms > 0:
            t_end = time.perf_counter() + args.settle_ms * 1e-3
            while time.perf_counter() < t_end:
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
            dt, kernel_ms_main, clk_main = timed_region()
        else:
            dt, kernel_ms_main, clk_main = dt_cold, km_cold, clk_cold
        per_rank = timed_region.per_rank
        # (kernel_ms_main: the dominant kernel, measured inside the timed region)
        # every kernel of a step, in an extra untimed pass of the same steps
        vqt.set_profiling(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        kernel_ms = vqt.last_kernel_ms()
        kernel_n = vqt.last_kernel_launches()
        kernel_ms.update(kernel_ms_main)
        r = dict(vqt=vqt, shard=shard, n_bins=n_bins, F=F, kernel_ms=kernel_ms, kernel_n=kernel_n, fpl=vqt.last_frames_per_launch(),
                 gemm_flop=vqt.last_gemm_flop(),     # flop the matrix instructions of one GEMM launch issue (tiles x 256 x 64 x depth x 2)
                 sclk_mhz=clk_main,                  # shader clock inside the GEMM kernel's K loop, sampled during the timed launches
                 dt_cold=dt_cold, km_cold=km_cold, clk_cold=clk_cold, per_rank=per_rank,
                 algo=vqt.last_algo())
        vqt.set_profiling(False)
        if with_alt:
            # the other GEMM arithmetic, same workload, reported beside the headline (rank-local, not part of `value`)
            other = "f32" if args.gemm == "bf16x3" else "bf16x3"
            vqt.set_gemm_precision(P.GEMM_F32 if other == "f32" else P.GEMM_BF16X3)
            step()
            torch.cuda.synchronize()
            vqt.set_profiling(True)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            r.update(other=other, dt_other=time.perf_counter() - t1, kernel_ms_other=vqt.last_kernel_ms())
            vqt.set_profiling(False)
            vqt.set_gemm_precision(P.GEMM_BF16X3 if args.gemm == "bf16x3" else P.GEMM_F32)
        r["dt"] = dt   # (max over ranks: timed_region)
        # sanity: the output is real (not skipped work)
        assert torch.isfinite(d_db).all() and float(d_db.max()) > 0.0 and int(d_cnt.sum()) > 0
        return r

    # The headline: BASELINE configs[1] (what BASELINE.json's metric is quoted on) at EVERY N, unless --config 2 asks otherwise
    main_cfg = args.config or 1
    W = WORKLOADS[main_cfg]
    R = measure(W, args.frames or W["frames"], with_alt=True)
    # ... and the other BASELINE bench configuration in a timed region of its own, for the extra key
    other_cfg = 2 if main_cfg == 1 else 1
    W2 = WORKLOADS[other_cfg]
    F2 = args.frames or W2["frames"]
    R2 = None if args.no_extra_config else measure(W2, F2, with_alt=False)
    vqt, N_BINS, F, dt = R["vqt"], R["n_bins"], R["F"], R["dt"]
    kernel_ms, kernel_n, fpl, gemm_flop, sclk_mhz = R["kernel_ms"], R["kernel_n"], R["fpl"], R["gemm_flop"], R["sclk_mhz"]
    other, dt_other, kernel_ms_other = R["other"], R["dt_other"], R["kernel_ms_other"]
    F_ALG_FLOP_PER_FRAME, B_ALG_BYTES_PER_FRAME = W["f_alg"], W["b_alg"]

    if rank == 0:
        total_frames = world * F * args.steps
        value = total_frames / dt
        # dominant kernel = largest total GPU time; one launch of it processes `fpl` frames
        dom = max(kernel_ms.items(), key=lambda kv: kv[1] * kernel_n.get(kv[0], 1))
        dom_s = dom[1] * 1e-3
        alg_tflops = F_ALG_FLOP_PER_FRAME * fpl / dom_s / 1e12
        gpu_ms_per_step = sum(kernel_ms[k] * kernel_n.get(k, 0) for k in kernel_ms) / args.steps
        split = args.gemm == "bf16x3" and vqt.last_algo() == P.ALGO_BLOCKDFT
        peak = PEAK_BF16_TFLOPS / 6.0 if split else PEAK_FP32_TFLOPS
        # roofline.achieved / frac: what the dominant kernel's matrix instructions EXECUTE per second (their flop per
        # launch, counted by the library from its tile list: tiles x 256 rows x 64 real columns x depth x 2, padding
        # columns and recomputed rows included) over the kernel's whole duration (K loop + tree + store), against the
        # dense fp32 MFMA peak.  It cannot exceed 1 and is reproducible from profiles/: SQ_INSTS_MFMA x 2048 (v_mfma_f32_16x16x4_f32) / kernel time.
        exec_tflops = None
        if dom[0] == "blockdft_gemm" and gemm_flop > 0:
            exec_tflops = gemm_flop / dom_s / 1e12
        achieved = exec_tflops if exec_tflops is not None else alg_tflops
        # roofline.traffic: HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (scripts/collect_profiles.sh writes
        # profiles/traffic_latest.json with the hash of the kernel sources it measured).  It is attached only when that hash is the
        # hash of the library running now — a capture of an older build is named, not used.
        traffic, traffic_source = None, None
        busy_frac, gui_cycles, counters_source = None, None, None   # the same capture's SQ counters: the matrix pipe's busy fraction and the kernel's active cycles (clock-independent)
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                now = open(os.path.join(ROOT, "pitchvis_amd", "lib", "libpvq.so.srchash")).read().strip()
                for tj in json.load(open(tpath)).get("captures", []):
                    if tj.get("kernel") == dom[0] and tj.get("frames_per_launch") == fpl and tj.get("n_bins") == N_BINS:
                        if tj.get("srchash") == now:
                            traffic = tj.get("hbm_bytes_per_launch")
                            traffic_source = f"{tj.get('file')} (kernel sources {now[:12]}, the running build)"
                            busy_frac, gui_cycles = tj.get("mfma_busy_frac"), tj.get("gui_active_cycles")
                            counters_source = f"{tj.get('sq_file')} (kernel sources {now[:12]}, the running build)" if busy_frac is not None else None
                        else:
                            traffic_source = f"stale: {tj.get('file')} measured kernel sources {str(tj.get('srchash'))[:12]}, running {now[:12]}"
            except Exception as e:   # a malformed capture file must not take the bench line down
                traffic_source = f"unreadable: {e}"
        out = {
            "metric": "vqt_frames_per_sec",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": W["name"].format(F=F, T=world * F, N=world),
                "frames_per_gpu_per_step": F,
                "hop": HOP,
                "n_bins": N_BINS,
                "algo": {P.ALGO_FFT: "fft", P.ALGO_BLOCKDFT: "blockdft"}.get(vqt.last_algo(), "auto"),
                "gemm_arith": ("fp32 operands as 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 per product block, fp32 accumulate "
                               "(error at fp32 rounding level, same parity bars)" if split else "fp32 MFMA v_mfma_f32_16x16x4_f32"),
                "sharding": f"one stream, frames x{world}, halo {vqt.window_union - HOP} samples per shard, no collective",
                # what the number was checked against: it travels with the headline
                "parity": "HIP path vs oracle/pvq_oracle.c (a CPU restatement of the Rust path; the Rust binary cannot be built here: "
                          "oracle unpinned vs the Rust binary); complex coefficients within 1e-5 of the frame peak, per-bin relative "
                          "1e-5 for bins within 20 dB of the frame peak; peak index sets bit-identical on identical dB frames; "
                          "find_peaks 0.1.5 distance-vs-prominence order unverified at 84 bins/octave (a no-op at this 36)",
            },
            "roofline": {
                # the binding roof of this path is fp32 matrix arithmetic (dense fp32 MFMA peak = fp32 vector peak);
                # the HBM view BASELINE.json asks for is given alongside (a few % of 8 TB/s).
                "bound": "mfma",
                "kernel": dom[0],
                "achieved": round(achieved, 3),
                "peak": round(peak, 1),
                "unit": "TFLOP/s",
                "frac": round(achieved / peak, 5),
                "traffic": traffic,
                "traffic_source": traffic_source,
                # rocprofv3 SQ counters of the same capture (scripts/collect_profiles.sh), in CYCLES — what rounds are compared in, whatever clock a box holds:
                # SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8), and GRBM_GUI_ACTIVE per launch (sum over the 8 XCDs)
                "mfma_busy_frac": busy_frac,
                "gui_active_cycles": gui_cycles,
                "counters_source": counters_source,
                "sclk_mhz": round(sclk_mhz, 1) if sclk_mhz else None,
                "executed_flop_per_launch": gemm_flop if exec_tflops is not None else None,
                # the FFT-route count of SURVEY 8d for the same frames, kept apart: the block-DFT path does not execute it
                "alg_flop_per_frame": F_ALG_FLOP_PER_FRAME,
                "alg_tflops": round(alg_tflops, 3),
                "alg_flop_ratio": round(F_ALG_FLOP_PER_FRAME * fpl / gemm_flop, 4) if exec_tflops is not None else None,
                "note": "achieved = flop issued by the matrix instructions of the kernel with the largest GPU time (the fused GEMM + "
                        "tree: its tiles x 256 rows x 64 real columns x hop/2 x 2) / its mean launch time, HIP events on the launch "
                        "stream; frac = achieved / the dense fp32 MFMA peak (157.3; 2500 / 6 fp32-equivalent for the split-bf16 "
                        "GEMM) = the fraction of the matrix pipe's time the kernel keeps it busy.  alg_tflops divides the FFT-route "
                        "flop count of SURVEY 8d (what the reference's algorithm would execute) by the same time; alg_flop_ratio = "
                        "that count / the executed one (> 1: a hop block is transformed once for all frames that share it, only the "
                        "columns the kernel reads, as two half-depth real GEMMs).  sclk_mhz: shader clock sampled inside the "
                        "kernel's K loop during the timed launches (the chip lowers it under MFMA load).  DESIGN.md 4-5",
                "frames_per_launch": fpl,
                "kernel_ms_per_launch": {k: round(v, 4) for k, v in kernel_ms.items()},
                "launches_per_step": {k: kernel_n.get(k, 0) // args.steps for k in kernel_ms},
                "gpu_ms_per_step_all_kernels": round(gpu_ms_per_step, 4),
                # whole path (all kernels of a step) against the FFT-route count: the algorithmic view, may exceed 1
                "path_alg_tflops": round(F_ALG_FLOP_PER_FRAME * F / (gpu_ms_per_step * 1e-3) / 1e12, 3),
                "hbm": {
                    "achieved": round(B_ALG_BYTES_PER_FRAME * F / (gpu_ms_per_step * 1e-3) / 1e9, 3),
                    "peak": PEAK_HBM_GBS,
                    "unit": "GB/s",
                    "frac": round(B_ALG_BYTES_PER_FRAME * F / (gpu_ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 6),
                    "alg_bytes_per_frame": B_ALG_BYTES_PER_FRAME,
                    "note": "algorithmic bytes of a step / GPU time of all its kernels",
                },
            },
        }
        out["alt_gemm"] = {
            "gemm": other,
            "value": round(F * args.steps / dt_other, 1),
            "unit": "frames/s (this rank)",
            "ms_per_step": round(dt_other / args.steps * 1e3, 4),
            "kernel_ms_per_launch": {k: round(v, 4) for k, v in kernel_ms_other.items()},
        }
        # the clock governor: what `value` waited for, and the window a cold device gives
        g_cold = R["km_cold"].get("blockdft_gemm")
        out["clock_settle"] = {
            "ms": args.settle_ms,
            "note": "untimed steps of the same workload before the W warm-up steps, so that the timed region sees the sustained shader clock "
                    "(an idle MI355X ramps over tens of ms of load; W + K steps of this workload are 12.5 ms): bench.py's docstring, DESIGN.md 5"}
        out["cold_start"] = {
            "value": round(total_frames / R["dt_cold"], 1),
            "unit": "frames/s",
            "ms_per_step": round(R["dt_cold"] / args.steps * 1e3, 4),
            "dominant_kernel_ms_per_launch": round(g_cold, 4) if g_cold else None,
            "roofline_frac": round(gemm_flop / (g_cold * 1e-3) / 1e12 / peak, 5) if g_cold and gemm_flop > 0 and dom[0] == "blockdft_gemm" else None,
            "sclk_mhz": round(R["clk_cold"], 1),
            "note": "the first timed window of the process, measured the same way (W warm-up steps from an idle device, then K steps between "
                    "barrier + synchronize, max over ranks); --settle-ms 0 makes it `value`",
        }
        if R2 is not None:
            # the other BASELINE bench configuration, same run, same K steps and barriers, its own timed region
            dom2 = max(R2["kernel_ms"].items(), key=lambda kv: kv[1] * R2["kernel_n"].get(kv[0], 1))
            exec2 = R2["gemm_flop"] / (dom2[1] * 1e-3) / 1e12 if dom2[0] == "blockdft_gemm" and R2["gemm_flop"] > 0 else None
            key = f"config{other_cfg}" + ("_single_gpu" if world == 1 else "")
            out[key] = {
                "workload": W2["name"].format(F=F2, T=world * F2, N=world),
                "value": round(world * F2 * args.steps / R2["dt"], 1),
                "unit": "frames/s",
                "n_gpus": world,
                "steps": args.steps,
                "ms_per_step": round(R2["dt"] / args.steps * 1e3, 4),
                "frames_per_gpu_per_step": F2,
                "n_bins": R2["n_bins"],
                "scaling": "weak",
                "kernel_ms_per_launch": {k: round(v, 4) for k, v in R2["kernel_ms"].items()},
                "roofline_frac": round(exec2 / PEAK_FP32_TFLOPS, 5) if exec2 is not None and args.gemm == "f32" else None,
                "note": "same run, its own timed region (barrier + synchronize on both sides, max over ranks); scaling efficiency of "
                        "this workload = its value at N / (N x its value at N = 1, reported as config2_single_gpu in the N = 1 line)",
            }
        if world > 1:
            # every rank's own view of the timed region (`value` uses the slowest): a slow rank shows here instead of hiding in the max
            out["per_rank"] = [{"rank": i, "ms_per_step": round(r_[0] / args.steps * 1e3, 4), "sclk_mhz": round(r_[1], 1),
                                "dominant_kernel_ms_per_launch": round(r_[2], 4)} for i, r_ in enumerate(R["per_rank"])]
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
