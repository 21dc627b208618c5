"""RCCL under this code, on the hardware that exists: ONE rank.  No multi-GPU node was available to the build
(DESIGN.md section 6), so the N > 1 path is covered by world-size-2 gloo tests on CPU (tests/test_dist_cpu.py,
tests/test_bench_cpu.py); this test makes sure that librccl itself has been loaded and has executed
mixdq_amd.shard's collectives on device tensors: the `nccl` backend (= RCCL on ROCm) at world size 1, the
bucketed broadcast of a quantized module's buffers, the max-over-ranks clock and the barrier.  Collected last:
it initialises (and destroys) the process-wide default process group."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rccl_world_size_one_runs_the_shard_collectives(C):
    import torch.distributed as dist
    from mixdq_amd import shard
    assert dist.is_available() and dist.is_nccl_available()
    assert not dist.is_initialized()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

        class Layer(torch.nn.Module):           # the buffer kinds of a QuantizedLinear / QuantizedConv2d
            def __init__(self):
                super().__init__()
                g = torch.Generator().manual_seed(5)
                self.register_buffer("weight_int", torch.randint(-128, 128, (1280, 1280), generator=g, dtype=torch.int8))
                self.register_buffer("scale", torch.rand(1280, generator=g))
                self.register_buffer("bias0", torch.rand(1280, generator=g))
                self.register_buffer("bias", torch.rand(1280, generator=g).half())
                self.register_buffer("act_zero_points", torch.tensor(-3.0))
        mod = torch.nn.Sequential(*[Layer() for _ in range(6)]).to(DEV)
        before = {k: v.clone() for k, v in mod.state_dict().items()}
        nbytes = shard.broadcast_module_state(mod, src=0, bucket_bytes=4 << 20)     # several buckets per dtype
        torch.cuda.synchronize()
        want = sum(v.numel() * v.element_size() for v in before.values())
        assert nbytes == want, (nbytes, want)
        for k, v in mod.state_dict().items():
            assert torch.equal(v, before[k]), k
        assert shard.max_over_ranks(12.5, DEV) == 12.5
        shard.barrier()
        # and the kernels still run on the device RCCL has been using
        x = torch.randn(64, 1280, device=DEV).half()
        q = C.quantize_per_tensor_to_int8(x, torch.tensor(10.0, device=DEV), torch.tensor(0.0, device=DEV))
        assert q.dtype == torch.int8 and q.shape == x.shape
    finally:
        dist.destroy_process_group()
