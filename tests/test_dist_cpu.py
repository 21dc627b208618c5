"""world_size-2 gloo tests of the N>1 path: batch sharding (no data-path collective) and the
one-time broadcast of rank 0's quantized buffers."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mixdq_amd import shard
    from tests.test_host import tiny_quantized, tiny_inputs
    r, lr, w = shard.init_distributed("gloo")
    assert (r, w) == (rank, world)
    unet, inp, _, _ = tiny_quantized()
    if rank != 0:   # make the replica differ, as if it had been calibrated on other data
        for b in unet.buffers():
            if b.dtype in (torch.float32, torch.float16):
                b.mul_(1.5)
            elif b.dtype == torch.int8:
                b.add_(1)
    nbytes = shard.broadcast_module_state(unet, src=0, bucket_bytes=1 << 16)
    import hashlib
    h = hashlib.sha256()
    for name, b in sorted(unet.named_buffers()):
        h.update(name.encode())
        h.update(b.contiguous().numpy().tobytes())
    batch = tiny_inputs(B=5)
    mine = shard.shard_batch(batch, rank, world)
    t = shard.max_over_ranks(float(rank + 1), "cpu")
    shard.barrier()
    q.put((rank, h.hexdigest(), nbytes, mine["sample"].shape[0],
           mine["added_cond_kwargs"]["text_embeds"].shape[0], mine["timestep"].dim(), t))


def test_broadcast_and_shard_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, h0, n0, b0, e0, d0, t0), (r1, h1, n1, b1, e1, d1, t1) = res
    assert h0 == h1, "replicas differ after the broadcast"
    assert n0 == n1 > 0
    assert (b0, b1) == (3, 2) and (e0, e1) == (3, 2) and d0 == d1 == 0   # 5 images -> 3 + 2
    assert t0 == t1 == 2.0


def test_shard_range_is_a_partition():
    from mixdq_amd.shard import shard_range
    for total in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_named_multi_gpu_config_shards_batch_64_evenly():
    """BASELINE.json configs[3]: global batch 64 over 1 / 2 / 4 / 8 ranks = 64 / 32 / 16 / 8 per GPU
    (bench.py --baseline-config 3: strong scaling)."""
    from mixdq_amd.shard import shard_range
    for world in (1, 2, 4, 8):
        sizes = [hi - lo for lo, hi in (shard_range(64, r, world) for r in range(world))]
        assert sizes == [64 // world] * world
