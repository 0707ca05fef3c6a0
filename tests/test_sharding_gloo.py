"""Multi-rank path on CPU: world_size-2 (and 3) gloo processes run the host-side plumbing bench.py uses at N > 1 —
the product's own host plan (pitchvis_amd.Vqt without a device: window union, bin count), pitchvis_amd.sharding
(plan_shard, global_stream with one seed on every rank, local_pcm with the real halo), the barrier and the
max-over-ranks reduce — and rank 0 checks that the gathered shards equal the unsharded computation bit for bit.
There is no GPU in this container and the product has no CPU fallback, so the frame transform itself is the CPU oracle
standing in for the kernels; the same shard plumbing through the HIP kernels is checked on the GPU box by
tests/test_configs_gpu.py::test_config3_sharded_equals_unsharded, ::test_config3_full_size_shard_on_one_gpu and
::test_bench_two_ranks_gloo_on_one_gpu."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pitchvis_amd.sharding import global_stream, local_pcm, plan_shard


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
    import pitchvis_amd as P
    op = O.OracleParams(sr=22050.0, min_freq=55.0, octaves=5, buckets_per_octave=36, quality=1.8, gamma=4.8 * 1.8)
    ov = O.OracleVqt(op)
    # the product's host plan (no device): the numbers bench.py shards with
    v = P.Vqt(P.VqtParameters(sr=22050.0, range=P.VqtRange(55.0, 5, 36), quality=1.8, gamma=4.8 * 1.8), device=None)
    union = v.window_union
    assert union == op.n_fft - min(ov.group_info(g)["window"][0] for g in range(ov.n_groups)) and v.n_bins == ov.n_bins
    stream = global_stream(0x5EED0003, hop * n_frames, "cpu")                       # same stream on every rank
    s = plan_shard(n_frames, hop, union, rank, world)
    mine = local_pcm(stream, s).numpy()
    pcm = stream.numpy()
    assert mine.size == s.n_lead + s.n_frames * hop and np.array_equal(mine, pcm[s.sample_begin:s.sample_end])
    local = ov.calculate_batch(mine, hop, s.n_frames, n_lead=s.n_lead)
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
