"""The REAL graph in the GPU suite (collected last): the 794-layer SDXL-Turbo UNet at 1024 px
(latent 128), with the reference's configs -- counterpart of the harness kernels/quantize_sdxl.py:
331-484, promoted from tools/check_full_unet.py.  Bit-level, not tolerance:

  * the fused graph (producer fusions, packed q|k|v, grouped k|v / time-embedding launches,
    GEMM+GEGLU, fused to_q + cross-attention, two-source GroupNorm, residual epilogues) equals its
    DE-FUSED reference (mixdq_amd.unet.defused: the same graph with every fused launch replaced by
    the chain of this repo's kernels it stands for -- FP16 output + the layer's own quantize
    launch, separate GEMMs, torch half adds, torch.cat) bit for bit: same arithmetic at every
    rounding point, so any difference is a wiring mistake at real shapes (row maps at T = 4096,
    the grouped launch with 70 members, 9 split shortcuts, 22 time_emb_proj);
  * hipGraph replay == eager;
  * row 0 of a batch-2 run == the batch-1 run of that image (per-image independence: what makes
    batch sharding over GPUs exact).

The FP16 PyTorch network is only run once, for calibration; no FP16 comparison legs."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Cfg:
    def __init__(self, w, a):
        self.w_config, self.a_config = w, a


def _slice_inputs(inp, lo, hi):
    return dict(sample=inp["sample"][lo:hi].contiguous(), timestep=inp["timestep"],
                encoder_hidden_states=inp["encoder_hidden_states"][lo:hi].contiguous(),
                added_cond_kwargs={k: v[lo:hi].contiguous()
                                   for k, v in inp["added_cond_kwargs"].items()})


def _check_graph(unet, inputs2, expect_accel, expect_w4=None):
    import mixdq_amd.unet as U
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from mixdq_amd.quantize_sdxl import hip_graph_opt
    unet.set_fused(True)
    qmods = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    assert len(qmods) == 794
    assert sum(m.valid_for_acceleration for m in qmods) == expect_accel
    if expect_w4 is not None:
        n4 = sum(m.valid_for_acceleration and getattr(m, "w_packed4", False) for m in qmods)
        assert n4 == expect_w4, n4
    inputs1 = _slice_inputs(inputs2, 0, 1)
    with torch.no_grad():
        fused2 = unet(**inputs2)[0].clone()
        again2 = unet(**inputs2)[0].clone()
        with U.defused():
            ref2 = unet(**inputs2)[0].clone()
        fused1 = unet(**inputs1)[0].clone()
        # ... and with every eligible LayerNorm riding in its producer GEMM's launch (mixdq_qlinear_w8a8_ln,
        # DESIGN.md 3.13; off by default): the same bits, 151 launches fewer at batch 1
        saved_chain = U.LN_CHAIN
        try:
            U.LN_CHAIN = True
            chain2 = unet(**inputs2)[0].clone()
            chain1 = unet(**inputs1)[0].clone()
        finally:
            U.LN_CHAIN = saved_chain
    assert torch.equal(chain2, fused2) and torch.equal(chain1, fused1), "LayerNorm-in-GEMM graph != the plain fused graph"
    assert torch.isfinite(fused2).all()
    assert torch.equal(fused2, again2), "fused graph is not deterministic"
    diff = (fused2.float() - ref2.float()).abs()
    assert torch.equal(fused2, ref2), (
        f"fused != de-fused reference: {int((diff > 0).sum())} of {diff.numel()} elements differ, "
        f"max {diff.max().item():.4g}")
    d1 = (fused2[:1].float() - fused1.float()).abs()
    assert torch.equal(fused2[:1], fused1), (
        f"row 0 of the batch-2 run != the batch-1 run: {int((d1 > 0).sum())} elements, "
        f"max {d1.max().item():.4g}")
    eager = unet.forward
    hip_graph_opt(unet)
    try:
        with torch.no_grad():
            g1 = unet(**inputs2)[0].clone()
            g2 = unet(**inputs2)[0].clone()
            # another batch size: a second graph; the first one must still replay correctly (its
            # persistent K/V and time-embedding buffers are keyed by shape, never dropped)
            h1 = unet(**inputs1)[0].clone()
            g3 = unet(**inputs2)[0].clone()
        assert len(unet.forward._cached) == 2
    finally:
        unet.forward = eager
    assert torch.equal(g1, fused2) and torch.equal(g2, fused2) and torch.equal(g3, fused2)
    assert torch.equal(h1, fused1)


def _build(w_name, a_name, w4_kernel):
    from mixdq_amd import cfgs
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import example_inputs, quantize_unet
    from mixdq_amd.unet import build_unet
    unet = build_unet(DEV)
    inputs2 = example_inputs(2, 128, DEV, seed=7)
    ckpt = calibrate(unet, [inputs2])
    bos = precompute_bos(unet, inputs2["encoder_hidden_states"])
    quantize_unet(unet, _Cfg(cfgs.load(w_name), cfgs.load(a_name)), ckpt, bos=True, bos_dict=bos,
                  w4_kernel=w4_kernel)
    del ckpt
    return unet, inputs2


def test_full_unet_w8a8_fused_equals_defused_graph_and_batch_rows(C):
    """uniform_8 + act_8.00 + BOS: 785 of 794 layers on the INT8 kernels (the bench configuration)."""
    unet, inputs2 = _build("weight/uniform_8", "act/act_8.00", False)
    _check_graph(unet, inputs2, expect_accel=785)
    del unet
    torch.cuda.empty_cache()


def test_full_unet_w4a8_mixed_fused_equals_defused_graph_and_batch_rows(C):
    """weight_4.00 + act_7.77 with the packed-W4 kernels (BASELINE.json configs[2]): 719 layers
    accelerated, 75 on the FP16 layer kernels."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    unet, inputs2 = _build("weight/weight_4.00", "act/act_7.77", True)
    qmods = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    n4_before = sum(m.valid_for_acceleration and getattr(m, "w_packed4", False) for m in qmods)
    assert n4_before == 485
    _check_graph(unet, inputs2, expect_accel=719)
    del unet
    torch.cuda.empty_cache()


def test_full_unet_w8a8_batch8_graph_fused_equals_defused_and_rows_equal_batch1(C):
    """The graph behind bench.py's `batch8` object and behind every rank of BASELINE.json configs[3] on 8 GPUs
    (64 images = 8 per GPU): at batch 8 the tile rule picks other kernels than at batch 1 / 2 (the four-phase
    256x256 tile, the 160-channel halo conv, un-fused to_q + cross-attention, prefetch payloads in a full
    attention launch).  Fused == de-fused bit for bit; hipGraph replay == eager; row 3 and row 7 equal the
    batch-1 runs of those images -- 8 images on one GPU and 1 image on each of 8 give the same tensors."""
    import mixdq_amd.unet as U
    from mixdq_amd.quantize_sdxl import example_inputs, hip_graph_opt
    unet, _ = _build("weight/uniform_8", "act/act_8.00", False)
    unet.set_fused(True)
    inputs8 = example_inputs(8, 128, DEV, seed=1042)
    with torch.no_grad():
        fused8 = unet(**inputs8)[0].clone()
        with U.defused():
            ref8 = unet(**inputs8)[0].clone()
        singles = {i: unet(**_slice_inputs(inputs8, i, i + 1))[0].clone() for i in (3, 7)}
    assert torch.isfinite(fused8).all()
    diff = (fused8.float() - ref8.float()).abs()
    assert torch.equal(fused8, ref8), (f"batch 8: fused != de-fused: {int((diff > 0).sum())} of "
                                       f"{diff.numel()} elements, max {diff.max().item():.4g}")
    for i, one in singles.items():
        assert torch.equal(fused8[i:i + 1], one), f"row {i} of the batch-8 run != its batch-1 run"
    eager = unet.forward
    hip_graph_opt(unet)
    try:
        with torch.no_grad():
            g1 = unet(**inputs8)[0].clone()
            g2 = unet(**inputs8)[0].clone()
    finally:
        unet.forward = eager
    assert torch.equal(g1, fused8) and torch.equal(g2, fused8)
    del unet
    torch.cuda.empty_cache()
