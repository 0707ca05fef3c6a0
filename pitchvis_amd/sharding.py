"""Frame sharding for multi-GPU runs: the hop batch splits into contiguous frame
ranges, one per rank, kernel tables replicated, no collective on the data path.  A shard's input
is its own hops plus a halo of preceding samples so that its first frame sees the same window the
unsharded stream would give it (the reference's rayon pattern gives each worker its own stream,
pitchvis_train/src/train.rs:146-155; here the workers share one stream)."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    first_frame: int   # global index of this rank's first frame
    n_frames: int
    sample_begin: int  # first sample of the global stream this rank must hold
    sample_end: int    # one past the last
    n_lead: int        # history samples preceding the shard's first hop inside [sample_begin, sample_end)


def plan_shard(n_frames_total: int, hop: int, window_union: int, rank: int, world: int) -> Shard:
    """Contiguous split; the first `n_frames_total % world` ranks take one extra frame.  The arithmetic lives in the library
    (pvq_plan_shard, pitchvis_amd/csrc/multi_host.cpp): the multi-device driver behind the C ABI and this Python face of it
    (torch.distributed ranks) cut a stream in the same places."""
    import ctypes as C
    from . import _lib
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    out = _lib.CShard()
    st = _lib.load().pvq_plan_shard(n_frames_total, hop, window_union, rank, world, C.byref(out))
    if st != 0:
        raise ValueError("bad rank/world")
    return Shard(int(out.first_frame), int(out.n_frames), int(out.sample_begin), int(out.sample_end), int(out.n_lead))


def stream_slice(seed: int, begin: int, end: int, device="cuda"):
    """Samples [begin, end) of the synthetic benchmark stream: white noise uniform in [-0.25, 0.25), fp32, a counter-based function of
    (seed, sample index) — splitmix64 of the index, its top 24 bits as the uniform — so that any rank computes exactly its own piece
    of ONE signal (halo included) without generating the rest, identically on every device type."""
    import torch
    i = torch.arange(begin, end, dtype=torch.int64, device=device)
    mask = lambda n: (1 << n) - 1   # noqa: E731  (logical right shifts on two's-complement int64)
    def to_i64(x):
        x &= (1 << 64) - 1
        return x - (1 << 64) if x >= (1 << 63) else x
    z = i * to_i64(0x9E3779B97F4A7C15) + to_i64((seed + 1) * 0xD1B54A32D192ED03)
    z = (z ^ ((z >> 30) & mask(34))) * to_i64(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & mask(37))) * to_i64(0x94D049BB133111EB)
    z = z ^ ((z >> 31) & mask(33))
    u = ((z >> 40) & mask(24)).to(torch.float32) * (1.0 / (1 << 24))   # [0, 1), exact in fp32
    return (u - 0.5) * 0.5


def global_stream(seed: int, n_samples: int, device="cuda"):
    """The whole synthetic benchmark stream (tests; the ranks of bench.py take stream_slice of their own piece)."""
    return stream_slice(seed, 0, n_samples, device)


def local_pcm(stream, shard: Shard):
    """This rank's piece of the global stream: its hops preceded by its halo (a copy, so the stream can be freed)."""
    return stream[shard.sample_begin:shard.sample_end].clone()
