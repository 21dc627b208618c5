"""UNet-level GPU tests (collected LAST, tests/conftest.py): module swap, hipGraph capture and the
fused graph on a small SDXL-shaped UNet.

These are tolerance tests of a whole network, so every bound is RELATIVE to a noise level measured
in the same test (the distance of the unfused W8A8 graph from the FP16 network), never a
hand-picked constant, and every quantizer state comes from a CPU FP32 copy of the (seeded)
network: PyTorch's FP16 GEMMs on the GPU are not bit-reproducible from run to run or box to box
(hipBLASLt picks its algorithm at run time), and scales that wobble in their last bits move the
INT8 rounding of thousands of activations.  Round 1's `test_fused_path_with_fp16_fallback_layers`
calibrated on the GPU FP16 network and compared with `0.05 * max + 0.02`; it missed that bound by
0.0007 on one box and passed on another.  tools/flake_probe.py (profiles/r02_flake_probe.txt): with
GPU calibration the max-error statistic wanders 0.038 .. 0.050 over 20 repetitions ON ONE BOX (20
distinct values) against a bound of 0.0535; with CPU calibration the fused-vs-unfused distance is
the same number in every repetition."""
import pytest
import torch

from tests.test_host import TINY, Args, tiny_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# the same small network with 64-wide heads: the fused graph then runs the HIP attention core
# (head_dim 64 only) instead of falling back to PyTorch's SDPA
TINY64 = dict(TINY, block_out_channels=(64, 128, 256), head_dim=64)


def _to_dev(host):
    return dict(sample=host["sample"].half().to(DEV), timestep=host["timestep"].to(DEV),
                encoder_hidden_states=host["encoder_hidden_states"].half().to(DEV),
                added_cond_kwargs={k: v.half().to(DEV) for k, v in host["added_cond_kwargs"].items()})


def _tiny_quantized_gpu(B=2, L=16, cfg=TINY, w_bits=None, a_drop=(), w4_kernel=False):
    """Tiny SDXL-shaped UNet, quantized on the GPU.  Calibration and the BOS rows come from a CPU
    FP32 copy of the same (seeded) network (see the module docstring).
    Returns (unet, inputs, FP16 network output on the GPU as fp32)."""
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    host = tiny_inputs(B=B, L=L)
    unet_c = build_unet("cpu", dtype=torch.float32, cfg=cfg)
    with torch.no_grad():
        ckpt = calibrate(unet_c, [host])
        bos = {k: v.half().to(DEV) for k, v in
               precompute_bos(unet_c, host["encoder_hidden_states"]).items()}
    del unet_c
    unet = build_unet(DEV, cfg=cfg)
    inp = _to_dev(host)
    with torch.no_grad():
        ref = unet(**inp)[0].float()
    names = list(quantizable_layers(unet))
    w = {"model." + n: (8 if w_bits is None else w_bits(i, n)) for i, n in enumerate(names)}
    a = {"model." + n: 8 for n in names if n not in a_drop}
    quantize_unet(unet, Args(w, a), ckpt, bos=True, bos_dict=bos, w4_kernel=w4_kernel)
    return unet, inp, ref


def _noise(out, ref):
    d = (out - ref).abs()
    return d.max().item(), d.mean().item()


def test_quantized_unet_runs_on_hip_kernels_and_tracks_fp16(C):
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    unet, inp, ref = _tiny_quantized_gpu()
    q = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    assert q and all(m.valid_for_acceleration for m in q)
    with torch.no_grad():
        out = unet(**inp)[0].float()
        again = unet(**inp)[0].float()
    assert torch.isfinite(out).all()
    assert torch.equal(out, again)
    # W8A8 with min-max scales on a random-weight network: the error is a few per cent of the
    # output's spread (a wiring mistake -- wrong scale, zero point, BOS row -- is O(spread))
    spread = ref.std().item()
    emax, emean = _noise(out, ref)
    assert emean < 0.05 * spread, (emean, spread)
    assert emax < 0.5 * spread, (emax, spread)


def test_quantized_unet_hip_graph_replay_is_bit_identical(C):
    from mixdq_amd.quantize_sdxl import hip_graph_opt
    unet, inp, _ = _tiny_quantized_gpu()
    with torch.no_grad():
        eager = unet(**inp)[0].clone()
    hip_graph_opt(unet)
    with torch.no_grad():
        g1 = unet(**inp)[0].clone()
        inp2 = dict(inp, sample=inp["sample"].flip(0).contiguous())
        g2 = unet(**inp2)[0].clone()
        eager2 = unet.forward.__wrapped__(**inp2)[0]
    assert torch.equal(eager, g1)
    assert torch.equal(eager2, g2)
    assert len(unet.forward._cached) == 1


def test_cuda_graph_opt_keeps_the_reference_signature(C):
    """quantize_sdxl.py:184: `cuda_graph_opt(unet, args)` -- a drop-in call passes the argparse
    namespace as the second positional argument."""
    from mixdq_amd.quantize_sdxl import cuda_graph_opt
    unet, inp, _ = _tiny_quantized_gpu(B=1)
    with torch.no_grad():
        eager = unet(**inp)[0].clone()
    ret = cuda_graph_opt(unet, Args(None, None))
    assert ret is unet
    with torch.no_grad():
        assert torch.equal(unet(**inp)[0], eager)


@pytest.mark.parametrize("cfg", [TINY, TINY64], ids=["heads16", "heads64"])
def test_fused_unet_matches_unfused_within_quantization_noise(C, cfg):
    """set_fused(True): producer fusions, residual epilogues, packed q|k|v and k|v GEMMs,
    GEMM+GEGLU, (heads64) the HIP attention core.  Same rounding points as the unfused graph;
    GroupNorm / SiLU / LayerNorm / GELU use this repo's arithmetic (within 1 FP16 ulp of PyTorch's
    FP32-reference ops, tests/test_fused_gpu.py) and PyTorch's own FP16 GroupNorm on ROCm differs
    from that reference in ~30 % of the elements, so the two graphs differ at quantization-noise
    level: fused must be no further from the FP16 network than unfused, and the two must agree to
    within that noise.  A wiring mistake is an O(1) error and fails this."""
    import mixdq_amd.unet as U
    unet, inp, ref = _tiny_quantized_gpu(cfg=cfg)
    with torch.no_grad():
        unfused = unet(**inp)[0].float()
        unet.set_fused(True)
        fused = unet(**inp)[0].float()
        again = unet(**inp)[0].float()
    assert torch.equal(fused, again)                      # deterministic
    blocks = [m for m in unet.modules() if isinstance(m, U.BasicTransformerBlock)]
    assert blocks and all(b._qkv_fused() is not None for b in blocks), "packed q|k|v GEMM not in use"
    nmax, nmean = _noise(unfused, ref)
    fmax, fmean = _noise(fused, ref)
    assert fmean <= 1.25 * nmean + 1e-3, (fmean, nmean)
    assert fmax <= 2.0 * nmax + 1e-3, (fmax, nmax)
    dmax, dmean = _noise(fused, unfused)
    assert dmean <= 2.0 * nmean + 1e-3, (dmean, nmean)


def test_fused_transformer_blocks_match_unfused_within_quantization_noise(C):
    """LayerNorm / GEGLU fusions and the residual epilogues against the unfused transformer blocks:
    same rounding points, but PyTorch's FP16 LayerNorm / GELU and the fused arithmetic may round an
    element differently, which flips an INT8 value now and then -- so the two graphs agree to well
    within the quantization noise (their distance from the FP16 network), not bit for bit."""
    import mixdq_amd.unet as U
    unet, inp, ref = _tiny_quantized_gpu()
    with torch.no_grad():
        unfused = unet(**inp)[0].float()
        for m in unet.modules():
            if type(m) is U.BasicTransformerBlock:
                m.fused = True
        fused = unet(**inp)[0].float()
        again = unet(**inp)[0].float()
    assert torch.equal(fused, again)
    nmax, nmean = _noise(unfused, ref)
    dmax, dmean = _noise(fused, unfused)
    assert dmax <= 1.5 * nmax + 1e-3, (dmax, nmax)
    assert dmean <= 1.0 * nmean + 1e-4, (dmean, nmean)


@pytest.mark.parametrize("cfg", [TINY, TINY64], ids=["heads16", "heads64"])
def test_fused_unet_graph_replay(C, cfg):
    from mixdq_amd.quantize_sdxl import hip_graph_opt
    unet, inp, _ = _tiny_quantized_gpu(cfg=cfg)
    unet.set_fused(True)
    with torch.no_grad():
        eager = unet(**inp)[0].clone()
    hip_graph_opt(unet)
    with torch.no_grad():
        g1 = unet(**inp)[0].clone()
        g2 = unet(**inp)[0].clone()
    assert torch.equal(eager, g1) and torch.equal(eager, g2)


def test_fused_fp16_forward_before_quantize_does_not_pin_the_unfused_launches(C):
    """bench.py's flow: an FP16 forward with set_fused(True) on the plain nn.Linear network, THEN
    quantize_unet.  The packed q|k|v / k|v decisions must be re-taken for the swapped layers
    (round 1 cached the negative answer for good: 210 extra launches per SDXL step)."""
    import mixdq_amd.unet as U
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    host = tiny_inputs(B=1, L=16)
    unet_c = build_unet("cpu", dtype=torch.float32, cfg=TINY64)
    with torch.no_grad():
        ckpt = calibrate(unet_c, [host])
        bos = {k: v.half().to(DEV) for k, v in
               precompute_bos(unet_c, host["encoder_hidden_states"]).items()}
    unet = build_unet(DEV, cfg=TINY64)
    inp = _to_dev(host)
    unet.set_fused(True)
    with torch.no_grad():
        unet(**inp)                               # FP16 layers on the fused glue
    unet.set_fused(False)
    names = list(quantizable_layers(unet))
    quantize_unet(unet, Args({"model." + n: 8 for n in names}, {"model." + n: 8 for n in names}),
                  ckpt, bos=True, bos_dict=bos)
    unet.set_fused(True)
    with torch.no_grad():
        unet(**inp)
    blocks = [m for m in unet.modules() if isinstance(m, U.BasicTransformerBlock)]
    assert all(b._qkv_fused() is not None for b in blocks)
    assert all(unet._kv_pack(b, inp["encoder_hidden_states"]) is not None for b in blocks)


def test_fused_caches_follow_in_place_buffer_updates(C):
    """load_state_dict / broadcast_module_state write new weights and scales into the SAME storage
    after a fused forward has built the packed q|k|v, k|v and border-table caches: the next forward
    must use the new values (round 1 kept the stale copies)."""
    unet, inp, _ = _tiny_quantized_gpu(B=1, cfg=TINY64)
    unet.set_fused(True)
    with torch.no_grad():
        before = unet(**inp)[0].clone()
        sd = {k: v.clone() for k, v in unet.state_dict().items()}
        # a different network of the same shapes: scale every per-channel epilogue vector
        changed = {k: (v * 1.25 if k.endswith((".scale", ".scale_0")) else v) for k, v in sd.items()}
        unet.load_state_dict(changed)
        mid = unet(**inp)[0].clone()
        unet.set_fused(False)
        mid_unfused = unet(**inp)[0].clone()
        unet.set_fused(True)
        unet.load_state_dict(sd)
        after = unet(**inp)[0].clone()
    assert torch.equal(before, after)
    assert not torch.equal(before, mid)
    # the fused graph saw the update everywhere the unfused graph did
    nmax, nmean = _noise(mid.float(), before.float())
    dmax, dmean = _noise(mid.float(), mid_unfused.float())
    assert dmean <= 0.25 * nmean + 1e-3, (dmean, nmean)


def test_fused_path_with_fp16_fallback_layers(C):
    """Activation-protected layers (no a_bit => FP16 fallback) inside fused blocks take the fp16
    output of the fused producer.  Compared with the UNFUSED graph that has the same fallbacks."""
    drop = {"conv_in", "conv_out", "down_blocks.0.resnets.0.conv2",
            "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.2",
            "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_k",
            "down_blocks.1.attentions.0.proj_in", "up_blocks.2.resnets.2.conv_shortcut"}
    for cfg in (TINY, TINY64):
        unet, inp, ref = _tiny_quantized_gpu(B=1, cfg=cfg, a_drop=drop)
        with torch.no_grad():
            unfused = unet(**inp)[0].float()
            unet.set_fused(True)
            fused = unet(**inp)[0].float()
        assert torch.isfinite(fused).all()
        nmax, nmean = _noise(unfused, ref)
        dmax, dmean = _noise(fused, unfused)
        # (tools/flake_probe.py, MI355X: d_mean / noise_mean = 0.81, d_max / noise_max = 0.78; a
        # wiring mistake is an error of the order of the output's spread, ~40x the noise)
        assert nmean < 0.1 * ref.std().item(), (nmean, ref.std().item())
        assert dmax <= 2.0 * nmax + 1e-3, (dmax, nmax)
        assert dmean <= 2.0 * nmean + 1e-4, (dmean, nmean)


def test_mixed_precision_unet_with_w4_kernels(C):
    """A mixed 8/4/2-bit weight config (the shape of the reference's weight_4.00.yaml): with
    w4_kernel=True every layer whose shape allows runs on the INT8 kernels.  The fused and unfused
    graphs agree far better than either agrees with FP16 (4-/2-bit weights on a random-weight
    network: large but bounded quantization noise)."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    unet, inp, ref = _tiny_quantized_gpu(
        B=1, w_bits=lambda i, n: ((8, 4, 8, 4, 2)[i % 5] if i % 10 else 2), w4_kernel=True)
    q = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    n4 = sum(m.valid_for_acceleration and m.w_packed4 for m in q)
    assert n4 > len(q) // 3
    outs = []
    for fused in (False, True):
        unet.set_fused(fused)
        with torch.no_grad():
            out = unet(**inp)[0].float()
        assert torch.isfinite(out).all()
        outs.append(out)
        assert (out - ref).abs().mean().item() < 2.0 * ref.std().item()
    nmax, nmean = _noise(outs[0], ref)
    dmax, dmean = _noise(outs[0], outs[1])
    assert dmean <= 0.5 * nmean + 1e-3, (dmean, nmean)
