"""Multi-rank path on CPU: world_size-2 (and 3) gloo processes each take a frame shard with its
halo, compute it (the CPU oracle stands in for the GPU kernel here), and rank 0 checks the
gathered result equals the unsharded computation bit for bit.  Exercises exactly the plumbing
bench.py uses at N > 1 (plan_shard, barrier, max-over-ranks reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pitchvis_amd.sharding import plan_shard


def test_plan_shard_covers_everything():
    for total, world in ((10, 3), (64, 2), (7, 8), (65536, 8), (1, 2)):
        seen = []
        for r in range(world):
            s = plan_shard(total, 256, 16384, r, world)
            seen += list(range(s.first_frame, s.first_frame + s.n_frames))
            assert s.sample_end - s.sample_begin == s.n_lead + s.n_frames * 256
            assert s.n_lead <= 16384 - 256
            if s.first_frame * 256 >= 16384 - 256:
                assert s.n_lead == 16384 - 256
        assert seen == list(range(total))


def _worker(rank, world, port, hop, n_frames, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    op = O.OracleParams(sr=22050.0, min_freq=55.0, octaves=5, buckets_per_octave=36, quality=1.8, gamma=4.8 * 1.8)
    ov = O.OracleVqt(op)
    union = op.n_fft - min(ov.group_info(g)["window"][0] for g in range(ov.n_groups))
    rng = np.random.default_rng(99)
    pcm = (rng.random(hop * n_frames, dtype=np.float32) - 0.5).astype(np.float32)  # same stream on every rank
    s = plan_shard(n_frames, hop, union, rank, world)
    local = ov.calculate_batch(pcm[s.sample_begin:s.sample_end], hop, s.n_frames, n_lead=s.n_lead)
    dist.barrier()
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduce of bench.py
    assert t.item() == float(world)
    counts = [plan_shard(n_frames, hop, union, r, world).n_frames for r in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad, ov.n_bins))
    buf[: s.n_frames] = torch.from_numpy(local)
    gathered = [torch.zeros((pad, ov.n_bins)) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank == 0:
        full = np.concatenate([g[:c].numpy() for g, c in zip(gathered, counts)], axis=0)
        want = ov.calculate_batch(pcm, hop, n_frames)
        q.put(bool(np.array_equal(full, want)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 256, 41, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
