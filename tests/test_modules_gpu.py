"""GPU parity tests at module level: QuantizedLinear / QuantizedConv2d (incl. BOS and split)
through the HIP kernels vs the outputs of the REFERENCE's classes (modules.npz, bit-exact) and
vs the reference's Path A fake-quant simulation (fakequant.npz, the reference's 1e-2 tolerance);
the UNet-level tests live in tests/test_unet_gpu.py (collected last)."""
import numpy as np
import pytest
import torch

from tests.cases import MODULE_CASES, module_ckpt, module_input
from tests.test_host import prepared

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["0", "1"], ids=["two_launches", "f16in"])
def f16in_mode(request, monkeypatch):
    """Every module test runs with the reference's two launches per layer (quantize, GEMM / conv) and with
    the quantize-in-prologue GEMM wherever it takes the shape (MIXDQ_F16IN=1; the default "auto" picks
    between the two by cost): the reference classes' outputs must come out of both, bit for bit."""
    import mixdq_amd._C as C_
    monkeypatch.setattr(C_, "F16IN", request.param)
    return request.param


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


def test_bos_layer_writes_into_a_callers_buffer_whose_row_0_is_prefilled(C, modules_golden):
    """forward(x, _bos_out=buf) (nn/glue.py _project_context: the swapped cross-attention keeps the buffer): same
    bits as forward(x), written into buf, row 0 left as the caller filled it; a buffer of another shape is ignored."""
    from mixdq_amd.nn.glue import _project_context
    c = next(c for c in MODULE_CASES if c["key"] == "lin_bos")
    qm = build(c, modules_golden)
    x = module_input(c).to(DEV)
    with torch.no_grad():
        want = qm(x)
        buf = torch.full_like(want, 7.0)
        buf[:, :1, :] = qm.bos_pre_computed
        got = qm(x, _bos_out=buf)
        assert got.data_ptr() == buf.data_ptr() and torch.equal(got, want)
        wrong = torch.empty((3,) + tuple(want.shape[1:]), dtype=want.dtype, device=DEV)
        y = qm(x, _bos_out=wrong)
        assert y.data_ptr() != wrong.data_ptr() and torch.equal(y, want)
        a, b = _project_context(qm, x), _project_context(qm, x.flip(1))        # one kept buffer per (B, T, device)
        assert a.data_ptr() == b.data_ptr() and torch.equal(b, qm(x.flip(1)))
        x2 = torch.cat([x, x.flip(1)], dim=0)
        assert _project_context(qm, x2).data_ptr() != a.data_ptr() and torch.equal(_project_context(qm, x2), qm(x2))
        assert len(qm.__dict__["_mixdq_bos_out"]) == 2
        qm.bos_pre_computed.mul_(2)                                            # the BOS row replaced in place: re-filled
        assert torch.equal(_project_context(qm, x), qm(x)) and _project_context(qm, x).data_ptr() == a.data_ptr()
        assert torch.equal(_project_context(qm, x.float()), qm(x.float()))     # not the kernel's input: the plain call
        # to_k / to_v of one cross-attention: equal quantizers share ONE quantize launch of the context
        from mixdq_amd.nn.glue import _project_kv
        qv = build(c, modules_golden)
        qv.weight_int.copy_(qv.weight_int.flip(0))
        k, v = _project_kv(qm, qv, x)
        assert torch.equal(k, qm(x)) and torch.equal(v, qv(x)) and k.data_ptr() == a.data_ptr() != v.data_ptr()
        n_shared = bench_count(lambda: _project_kv(qm, qv, x))
        qv.act_scales_inv.mul_(1.5)                                            # unequal quantizers: each its own
        qv.act_scales.div_(1.5)
        k, v = _project_kv(qm, qv, x)
        assert torch.equal(k, qm(x)) and torch.equal(v, qv(x))
        # (where the quantize-in-GEMM launch is preferred -- MIXDQ_F16IN=1 -- neither form has a quantize launch)
        fused = C.qlinear_f16in_wanted(x, qm.out_features, qm.in_features, bos=True)
        assert bench_count(lambda: _project_kv(qm, qv, x)) == n_shared + (0 if fused else 1)


def bench_count(fn):
    import bench
    return bench.count_kernels(fn, torch.device(DEV))


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
