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
    """Contiguous split; the first `n_frames_total % world` ranks take one extra frame."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames_total, world)
    n = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    hop_begin = first * hop                      # first new sample of this shard
    halo = max(window_union - hop, 0)            # samples before hop_begin its first frame reads
    begin = max(hop_begin - halo, 0)
    return Shard(first, n, begin, hop_begin + n * hop, hop_begin - begin)


def global_stream(seed: int, n_samples: int, device="cuda"):
    """The synthetic benchmark stream: white noise uniform in [-0.25, 0.25), fp32.  Same seed => same values on every
    rank (one Philox stream per device type), so that ranks slicing it hold consecutive pieces of ONE signal."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (torch.rand(n_samples, device=device, generator=g) - 0.5) * 0.5


def local_pcm(stream, shard: Shard):
    """This rank's piece of the global stream: its hops preceded by its halo (a copy, so the stream can be freed)."""
    return stream[shard.sample_begin:shard.sample_end].clone()
