"""GPU parity tests at module level: QuantizedLinear / QuantizedConv2d (incl. BOS and split)
through the HIP kernels vs the outputs of the REFERENCE's classes (modules.npz, bit-exact) and
vs the reference's Path A fake-quant simulation (fakequant.npz, the reference's 1e-2 tolerance);
module swap + hipGraph capture of a small UNet."""
import numpy as np
import pytest
import torch

from tests.cases import MODULE_CASES, module_ckpt, module_input
from tests.test_host import prepared, tiny_inputs, TINY, Args

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(c, golden):
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    qm = cls.from_float(prepared(c, golden), split=c.get("split", 0), ckpt=module_ckpt(c, golden))
    return qm.to(DEV)


@pytest.mark.parametrize("c", MODULE_CASES, ids=[c["key"] for c in MODULE_CASES])
def test_module_forward_bit_exact_vs_reference_classes(C, modules_golden, c):
    qm = build(c, modules_golden)
    assert qm.valid_for_acceleration
    x = module_input(c).to(DEV)
    if c["kind"] == "conv":
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = qm(x)
    want = modules_golden[f"{c['key']}.out"]
    got = y.contiguous().cpu().numpy()
    assert got.shape == want.shape and y.dtype == torch.float16
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), \
        f"{(got.view(np.uint16) != want.view(np.uint16)).sum()} of {got.size} differ"
    if c["kind"] == "conv":
        assert y.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("c", [c for c in MODULE_CASES if c["kind"] == "conv"],
                         ids=[c["key"] for c in MODULE_CASES if c["kind"] == "conv"])
def test_conv_module_accepts_nchw_input(C, modules_golden, c):
    """NCHW-contiguous activations (what the reference's UNet feeds) give the same result."""
    qm = build(c, modules_golden)
    with torch.no_grad():
        y = qm(module_input(c).to(DEV).contiguous())
    want = modules_golden[f"{c['key']}.out"]
    assert np.array_equal(y.contiguous().cpu().numpy().view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("c", MODULE_CASES, ids=[c["key"] for c in MODULE_CASES])
def test_module_within_reference_tolerance_of_fake_quant(C, modules_golden, fakequant_golden, c):
    """INT8 kernels vs the qdiff simulation (Path A).
    (1) Path A restated (oracle/fakequant.py, pinned bit-for-bit to the reference's QuantLayer by
        tests/test_oracle.py) evaluated with the SAME fp16-rounded scales the kernels use:
        rtol = atol = 1e-2, the reference's own int-vs-fp tolerance (op/qlinear.py:101,
        op/qconv2d.py:100).
    (2) The reference QuantLayer's stored output, which used the un-rounded fp32 scales
        (convert_ckpt.py:36 rounds them to fp16 afterwards): max error within 1e-2 + 0.5 % of
        the output range."""
    from oracle.fakequant import quant_layer_forward
    from tests.cases import make_float_module
    qm = build(c, modules_golden)
    x = module_input(c)
    with torch.no_grad():
        y = qm(x.to(DEV)).float().cpu()
    key, split = c["key"], c.get("split", 0)

    def ck(sfx, field):
        return torch.from_numpy(modules_golden[f"{key}.ckpt.{sfx}.{field}"]).float()[2]

    fm = make_float_module(c).half().float()     # the kernels quantize the fp16 weights
    kw = None
    if c["kind"] == "conv":
        kw = dict(stride=fm.stride, padding=fm.padding, dilation=fm.dilation, groups=fm.groups)
    extra = ()
    if split:
        extra = (split, ck("weight_quantizer_0", "delta_list"), ck("act_quantizer_0", "delta_list"),
                 ck("act_quantizer_0", "zero_point_list"))
    with torch.no_grad():
        sim = quant_layer_forward(x.float(), fm.weight, fm.bias, ck("weight_quantizer", "delta_list"),
                                  ck("act_quantizer", "delta_list"),
                                  ck("act_quantizer", "zero_point_list"), 8, 8, kw, *extra)
    ref = torch.from_numpy(fakequant_golden[f"{key}.pathA_w8a8"])
    if c.get("bos"):     # the simulation has no layer-level BOS splice: compare tokens 1..
        y, sim, ref = y[:, 1:], sim[:, 1:], ref[:, 1:]
    torch.testing.assert_close(y, sim, rtol=1e-2, atol=1e-2)
    assert (y - ref).abs().max().item() <= 1e-2 + 5e-3 * ref.abs().max().item()


def test_bos_batch2(C, modules_golden):
    c = next(c for c in MODULE_CASES if c["key"] == "lin_bos")
    qm = build(c, modules_golden)
    x = module_input(c).to(DEV)
    x2 = torch.cat([x, x.flip(1)], dim=0)
    with torch.no_grad():
        y2, ya, yb = qm(x2), qm(x), qm(x.flip(1))
    assert torch.equal(y2[0], ya[0]) and torch.equal(y2[1], yb[0])
    assert torch.equal(y2[:, 0], qm.bos_pre_computed.expand(2, -1, -1)[:, 0])


def test_split_shortcut_batch2_channels_last_and_nchw(C, modules_golden):
    c = next(c for c in MODULE_CASES if c["key"] == "conv_split")
    qm = build(c, modules_golden)
    x = module_input(c).to(DEV)
    x2 = torch.cat([x, x.flip(2)], dim=0)
    with torch.no_grad():
        y_nchw = qm(x2.contiguous())
        y_cl = qm(x2.contiguous(memory_format=torch.channels_last))
        y1 = qm(x)
    assert torch.equal(y_nchw, y_cl)
    assert torch.equal(y_nchw[:1], y1)


def test_non_fp16_input_uses_dequantised_weight_fallback(C, modules_golden):
    c = MODULE_CASES[0]
    qm = build(c, modules_golden)
    x = module_input(c).to(DEV).float()
    with torch.no_grad():
        y = qm(x)
    w = qm.weight_int.float() * qm.weight_scales[:, None]
    torch.testing.assert_close(y, torch.nn.functional.linear(x, w, qm.bias.float()))


def _tiny_quantized_gpu():
    """Tiny SDXL-shaped UNet, quantized W8A8 on the GPU.  Calibration and the BOS rows come from a
    CPU FP32 copy of the same (seeded) network: PyTorch's FP16 GEMMs on the GPU are not bit-
    reproducible from run to run, and scales that wobble in their last bits would make the exact-
    wiring checks below depend on rounding luck."""
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    host = tiny_inputs(B=2, L=16)
    unet_c = build_unet("cpu", dtype=torch.float32, cfg=TINY)
    with torch.no_grad():
        ckpt = calibrate(unet_c, [host])
        bos = {k: v.half().to(DEV) for k, v in
               precompute_bos(unet_c, host["encoder_hidden_states"]).items()}
    del unet_c
    unet = build_unet(DEV, cfg=TINY)
    inp = dict(sample=host["sample"].half().to(DEV), timestep=host["timestep"].to(DEV),
               encoder_hidden_states=host["encoder_hidden_states"].half().to(DEV),
               added_cond_kwargs={k: v.half().to(DEV) for k, v in host["added_cond_kwargs"].items()})
    with torch.no_grad():
        ref = unet(**inp)[0].float()
    names = list(quantizable_layers(unet))
    quantize_unet(unet, Args({"model." + n: 8 for n in names}, {"model." + n: 8 for n in names}),
                  ckpt, bos=True, bos_dict=bos)
    return unet, inp, ref


def test_quantized_unet_runs_on_hip_kernels_and_tracks_fp16(C):
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    unet, inp, ref = _tiny_quantized_gpu()
    q = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    assert q and all(m.valid_for_acceleration for m in q)
    with torch.no_grad():
        out = unet(**inp)[0].float()
    assert torch.isfinite(out).all()
    err = (out - ref).abs().max().item()
    assert err < 0.05 * ref.abs().max().item() + 0.02, err


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


def test_fused_unet_tracks_fp16_at_least_as_well_as_unfused(C):
    """set_fused(True): producer fusions + residual epilogues.  Same rounding points as the
    unfused graph; GroupNorm / SiLU / LayerNorm / GELU use this repo's arithmetic (within 1 FP16
    ulp of PyTorch's FP32-reference ops, tests/test_fused_gpu.py).  PyTorch's own FP16 GroupNorm on
    ROCm keeps its fused scale/shift in FP16 and differs from that reference in ~30 % of the
    elements, so fused and unfused graphs differ at quantization-noise level; what must hold is
    that the fused graph is no further from the FP16 network than the unfused one."""
    unet, inp, ref_fp16 = _tiny_quantized_gpu()
    with torch.no_grad():
        unfused = unet(**inp)[0].float()
        unet.set_fused(True)
        fused = unet(**inp)[0].float()
        again = unet(**inp)[0].float()
    assert torch.equal(fused, again)                      # deterministic
    e_unf = (unfused - ref_fp16).abs().mean().item()
    e_fus = (fused - ref_fp16).abs().mean().item()
    assert e_fus <= 1.25 * e_unf + 1e-3, (e_fus, e_unf)
    assert (fused - ref_fp16).abs().max().item() < 0.05 * ref_fp16.abs().max().item() + 0.02
    assert (fused - unfused).abs().mean().item() <= 2.0 * e_unf + 1e-3


def test_fused_transformer_blocks_match_unfused_within_quantization_noise(C):
    """LayerNorm / GEGLU fusions and the residual epilogues against the unfused transformer blocks:
    same rounding points, but PyTorch's FP16 LayerNorm / GELU and the fused arithmetic may round an
    element differently, which flips an INT8 value now and then -- so the two graphs agree to well
    within the quantization noise (their distance from the FP16 network), not bit for bit.  A
    wiring mistake (wrong quantizer, BOS row, residual) is an O(1) error and fails this."""
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
    noise_max = (unfused - ref).abs().max().item()
    noise_mean = (unfused - ref).abs().mean().item()
    d = (fused - unfused).abs()
    assert d.max().item() <= noise_max + 1e-3, (d.max().item(), noise_max)
    assert d.mean().item() <= 0.5 * noise_mean + 1e-4, (d.mean().item(), noise_mean)


def test_fused_unet_graph_replay(C):
    from mixdq_amd.quantize_sdxl import hip_graph_opt
    unet, inp, _ = _tiny_quantized_gpu()
    unet.set_fused(True)
    with torch.no_grad():
        eager = unet(**inp)[0].clone()
    hip_graph_opt(unet)
    with torch.no_grad():
        g1 = unet(**inp)[0].clone()
    assert torch.equal(eager, g1)


def test_fused_path_with_fp16_fallback_layers(C):
    """Activation-protected layers (no a_bit => FP16 fallback) inside fused blocks take the fp16
    output of the fused producer."""
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    unet = build_unet(DEV, cfg=TINY)
    inp = tiny_inputs(B=1, L=16)
    inp = dict(sample=inp["sample"].half().to(DEV), timestep=inp["timestep"].to(DEV),
               encoder_hidden_states=inp["encoder_hidden_states"].half().to(DEV),
               added_cond_kwargs={k: v.half().to(DEV) for k, v in inp["added_cond_kwargs"].items()})
    with torch.no_grad():
        ref = unet(**inp)[0].float()
    ckpt = calibrate(unet, [inp])
    bos = precompute_bos(unet, inp["encoder_hidden_states"])
    names = list(quantizable_layers(unet))
    drop = {"conv_in", "conv_out", "down_blocks.0.resnets.0.conv2",
            "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.2",
            "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_k",
            "down_blocks.1.attentions.0.proj_in"}
    quantize_unet(unet, Args({"model." + n: 8 for n in names},
                             {"model." + n: 8 for n in names if n not in drop}),
                  ckpt, bos=True, bos_dict=bos)
    unet.set_fused(True)
    with torch.no_grad():
        out = unet(**inp)[0].float()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < 0.05 * ref.abs().max().item() + 0.02


# ----------------------------------------------------------------------------- W4A8 (f-2)
@pytest.mark.parametrize("c", [c for c in MODULE_CASES if c["cin"] % 32 == 0 and not c.get("split")],
                         ids=[c["key"] for c in MODULE_CASES if c["cin"] % 32 == 0 and not c.get("split")])
def test_w4a8_module_on_hip_kernels_tracks_path_a(C, modules_golden, fakequant_golden, c):
    """w4_kernel=True: the layer runs the packed-W4 INT8 kernels (the reference would fall back to
    FP16).  Oracle = Path A at 4-bit weights: restated with the same fp16
    checkpoint scales, rtol = atol = 1e-2."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from oracle.fakequant import quant_layer_forward
    from tests.cases import make_float_module
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    fm = prepared(c, modules_golden, w_bit=4)
    fm.w4_kernel = True
    qm = cls.from_float(fm, ckpt=module_ckpt(c, modules_golden)).to(DEV)
    assert qm.valid_for_acceleration and qm.w_packed4
    x = module_input(c)
    with torch.no_grad():
        y = qm(x.to(DEV)).float().cpu()
    key = c["key"]

    def ck(sfx, field, i):
        return torch.from_numpy(modules_golden[f"{key}.ckpt.{sfx}.{field}"]).float()[i]

    fm32 = make_float_module(c).half().float()
    kw = None
    if c["kind"] == "conv":
        kw = dict(stride=fm32.stride, padding=fm32.padding, dilation=fm32.dilation, groups=1)
    with torch.no_grad():
        sim = quant_layer_forward(x.float(), fm32.weight, fm32.bias,
                                  ck("weight_quantizer", "delta_list", 1),
                                  ck("act_quantizer", "delta_list", 2),
                                  ck("act_quantizer", "zero_point_list", 2), 4, 8, kw)
    if c.get("bos"):
        y, sim = y[:, 1:], sim[:, 1:]
    torch.testing.assert_close(y, sim, rtol=1e-2, atol=1e-2)


def test_mixed_precision_unet_with_w4_kernels(C):
    """A mixed 8/4/2-bit weight config (the shape of the reference's weight_4.00.yaml): with
    w4_kernel=True every layer whose shape allows runs on the INT8 kernels, and the network tracks
    its FP16 version at 4-bit quantization-noise level."""
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import build_unet, quantizable_layers
    unet = build_unet(DEV, cfg=TINY)
    inp = tiny_inputs(B=1, L=16)
    inp = dict(sample=inp["sample"].half().to(DEV), timestep=inp["timestep"].to(DEV),
               encoder_hidden_states=inp["encoder_hidden_states"].half().to(DEV),
               added_cond_kwargs={k: v.half().to(DEV) for k, v in inp["added_cond_kwargs"].items()})
    with torch.no_grad():
        ref = unet(**inp)[0].float()
    ckpt = calibrate(unet, [inp])
    bos = precompute_bos(unet, inp["encoder_hidden_states"])
    names = list(quantizable_layers(unet))
    bits = {"model." + n: (8, 4, 8, 4, 2)[i % 5] if i % 10 else 2 for i, n in enumerate(names)}
    quantize_unet(unet, Args(bits, {"model." + n: 8 for n in names}), ckpt, bos=True, bos_dict=bos,
                  w4_kernel=True)
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
        # 4-/2-bit weights on a random-weight network: large but bounded quantization noise
        assert (out - ref).abs().mean().item() < ref.abs().mean().item() + 0.02
    # the fused and unfused graphs agree far better than either agrees with FP16
    assert (outs[0] - outs[1]).abs().mean().item() < 0.5 * (outs[0] - ref).abs().mean().item() + 0.01
