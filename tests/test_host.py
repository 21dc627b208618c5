"""CPU tests of the host logic: QuantizedLinear / QuantizedConv2d buffer derivation against the
buffers the REFERENCE's own from_float produced (modules.npz), forward dispatch (BOS, split,
fallbacks) with the operators stubbed by the oracle, module swap, yaml registration, configs,
calibration format."""
import numpy as np
import pytest
import torch
import torch.nn as nn
from torch.ao.quantization import PlaceholderObserver, QConfig

from tests.cases import MODULE_CASES, make_float_module, module_ckpt, module_input


def prepared(c, golden, w_bit=8, a_bit=8, with_act=True):
    fm = make_float_module(c).half()
    fm.module_name = c["name"]
    wd = {8: torch.qint8, 4: torch.quint4x2, 2: torch.quint4x2}[w_bit]
    fm.qconfig = QConfig(activation=PlaceholderObserver.with_args(dtype=torch.qint8),
                         weight=PlaceholderObserver.with_args(dtype=wd))
    fm.w_bit = w_bit
    if with_act:
        fm.a_bit = a_bit
    if c.get("bos"):
        fm.bos = True
        fm.bos_pre_computed = torch.from_numpy(golden[f"{c['key']}.bos_pre_computed"])
    return fm


@pytest.fixture
def oracle_ops(monkeypatch, oracle):
    """Stand-ins for the three HIP operators, backed by the oracle, so forward() dispatch can be
    exercised on a CPU-only box.  Test-only: the product has no such fallback."""
    import mixdq_amd.nn.Linear as L
    import mixdq_amd.nn.Conv2d as Cv

    def quant(x, s_inv, zp):
        out = torch.empty_like(x, dtype=torch.int8)
        out.copy_(torch.from_numpy(oracle.quantize(x.numpy(), float(s_inv), float(zp))))
        return out

    def qlin(x_int, w, ws, a_s, a_zp, wsum, scale, bias0, bias=None, _out=None, _row_map=None,
             _residual=None, _residual_div=1, _cfg=0, _w4=False):
        if _w4:
            w = torch.from_numpy(oracle.unpack_w4(w.contiguous().numpy()))
        D = torch.from_numpy(oracle.qlinear(x_int.contiguous().numpy(), w.numpy(), bias0.numpy(),
                                            scale.numpy(), None if bias is None else bias.numpy()))
        if _residual is not None:
            D = D + _residual.reshape(D.shape)
        if _out is None:
            return D
        g, stride, off = _row_map
        _out.view(-1, _out.shape[-1]).view(-1, stride, _out.shape[-1])[:, off:off + g] = \
            D.reshape(-1, g, D.shape[-1])
        return _out

    def qconv(x_int, w, ws, a_s, a_zp, scale, wsum, bias0, bias=None, stride=1, padding=0,
              dilation=1, _table=None, _residual=None, _residual_per_image=False, _cfg=0,
              _w4=False, _upsample2x=False):
        assert not _upsample2x
        wk = w.permute(0, 2, 3, 1).contiguous().numpy()
        if _w4:
            wk = oracle.unpack_w4(wk)
        D = oracle.qconv2d(x_int.permute(0, 2, 3, 1).contiguous().numpy(), wk, scale.numpy(),
                           None if wsum is None else wsum.numpy(), float(a_zp),
                           None if bias0 is None else bias0.numpy(),
                           None if bias is None else bias.numpy(), stride, padding)
        out = torch.from_numpy(D).permute(0, 3, 1, 2)
        if _residual is not None:
            out = out + (_residual[:, :, None, None] if _residual_per_image else _residual)
        return out

    monkeypatch.setattr(L, "quant_op", quant)
    monkeypatch.setattr(L, "qlinear", qlin)
    monkeypatch.setattr(Cv, "quant_op", quant)
    monkeypatch.setattr(Cv._C, "qconv2d_w8_a8_ohalf", qconv)
    monkeypatch.setattr(Cv.QuantizedConv2d, "_border_table", lambda self, sfx: None)


@pytest.mark.parametrize("c", MODULE_CASES, ids=[c["key"] for c in MODULE_CASES])
def test_from_float_buffers_match_reference(modules_golden, c):
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    import hashlib
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    qm = cls.from_float(prepared(c, modules_golden), split=c.get("split", 0),
                        ckpt=module_ckpt(c, modules_golden))
    assert qm.valid_for_acceleration
    ref = {k[len(c["key"]) + 5:]: v for k, v in modules_golden.items()
           if k.startswith(c["key"] + ".buf.")}
    refsha = {k[len(c["key"]) + 8:]: v for k, v in modules_golden.items()
              if k.startswith(c["key"] + ".bufsha.")}
    mine = dict(qm.named_buffers())
    assert set(mine) == set(ref) | set(refsha), (sorted(mine), sorted(ref), sorted(refsha))
    for name, want in ref.items():
        got = mine[name].numpy()
        assert got.dtype == want.dtype and got.shape == want.shape, name
        assert np.array_equal(got, want), name
    for name, want in refsha.items():
        got = hashlib.sha256(mine[name].contiguous().numpy().tobytes()).hexdigest()
        assert got == str(want), name
    # attributes the reference sets to None for the unused epilogue form (nn/Conv2d.py:171,177)
    if c["kind"] == "conv":
        if c["pad"] == 0:
            assert qm.weight_sum_by_input_channels is None
        else:
            assert qm.bias0 is None


@pytest.mark.parametrize("c", MODULE_CASES, ids=[c["key"] for c in MODULE_CASES])
def test_forward_dispatch_matches_reference_output(modules_golden, oracle_ops, c):
    """forward() over oracle-backed ops == the reference class's forward over the same ops."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    qm = cls.from_float(prepared(c, modules_golden), split=c.get("split", 0),
                        ckpt=module_ckpt(c, modules_golden))
    with torch.no_grad():
        y = qm(module_input(c))
    want = modules_golden[f"{c['key']}.out"]
    assert tuple(y.shape) == want.shape
    assert np.array_equal(y.contiguous().numpy().view(np.uint16), want.view(np.uint16))


def test_bos_forward_batch2_uses_strided_semantics(modules_golden, oracle_ops):
    """At batch > 1 the reference quantizes x[:,1:,:] linearly (wrong tokens); this build uses the
    slice's strides: each batch element equals its own batch-1 result."""
    from mixdq_amd.nn import QuantizedLinear
    c = next(c for c in MODULE_CASES if c["key"] == "lin_bos")
    qm = QuantizedLinear.from_float(prepared(c, modules_golden), ckpt=module_ckpt(c, modules_golden))
    x = module_input(c)
    x2 = torch.cat([x, x.flip(1)], dim=0)
    with torch.no_grad():
        y2 = qm(x2)
        y_a, y_b = qm(x), qm(x.flip(1))
    assert torch.equal(y2[0], y_a[0]) and torch.equal(y2[1], y_b[0])
    assert torch.equal(y2[:, 0], qm.bos_pre_computed.expand(2, -1, -1)[:, 0])


@pytest.mark.parametrize("c", MODULE_CASES[:1] + MODULE_CASES[3:4], ids=["linear", "conv"])
def test_fallback_rules(modules_golden, c):
    """4-/2-bit weights, missing a_bit and misaligned sizes fall back to FP16 F.linear / F.conv2d
    on the un-quantized weight (nn/Linear.py:31,37-43,133-134,155-156)."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    ck = module_ckpt(c, modules_golden)
    for kw in (dict(w_bit=4), dict(w_bit=2), dict(with_act=False)):
        qm = cls.from_float(prepared(c, modules_golden, **kw), ckpt=ck)
        assert not qm.valid_for_acceleration
        assert "weight" in dict(qm.named_buffers()) and "weight_int" not in dict(qm.named_buffers())
        assert "Fallback" in qm._get_name()
        x = module_input(c).float()
        qm = qm.float()
        ref = make_float_module(c).half().float()
        with torch.no_grad():
            torch.testing.assert_close(qm(x), ref(x))


def test_misaligned_linear_falls_back():
    from mixdq_amd.nn import QuantizedLinear
    from mixdq_amd.nn.utils import QParam
    w = QParam(torch.per_channel_affine, torch.qint8, torch.ones(6), torch.zeros(6), 0)
    a = QParam(torch.per_tensor_affine, torch.qint8, torch.tensor(0.1), torch.tensor(3.0), 0)
    assert QuantizedLinear(8, 6, w_qparams=w, a_qparams=a).valid_for_acceleration is False
    assert QuantizedLinear(8, 8, w_qparams=QParam(torch.per_channel_affine, torch.qint8,
                                                  torch.ones(8), torch.zeros(8), 0),
                           a_qparams=a).valid_for_acceleration is True
    wz = QParam(torch.per_channel_affine, torch.qint8, torch.ones(8), torch.ones(8), 0)
    assert QuantizedLinear(8, 8, w_qparams=wz, a_qparams=a).valid_for_acceleration is False


def test_get_quant_para_and_uint4_helpers():
    from mixdq_amd.nn.utils import (get_quant_para, quantize_per_tensor_uint4, unpack_uint4,
                                    dequantize_per_tensor_uint4, bit_index)
    assert [bit_index(b) for b in (2, 4, 8)] == [0, 1, 2]
    ck = {"l.weight_quantizer": dict(delta_list=torch.arange(6.).reshape(3, 2),
                                     zero_point_list=torch.zeros(3, 2)),
          "l.act_quantizer": dict(delta_list=torch.tensor([.4, .2, .1]),
                                  zero_point_list=torch.tensor([1., 7., 130.])),
          "l.weight_quantizer_0": dict(delta_list=torch.ones(3, 2), zero_point_list=torch.zeros(3, 2)),
          "l.act_quantizer_0": dict(delta_list=torch.tensor([.4, .2, .3]),
                                    zero_point_list=torch.tensor([1., 7., 128.]))}
    s, z, s0, z0 = get_quant_para(ck, 8, "l", "act")
    assert float(s) == pytest.approx(0.1) and float(z) == 2.0 and s0 is None   # 130 - 128
    s, z, s0, z0 = get_quant_para(ck, 4, "l", "weight", split=3)
    assert s.tolist() == [2., 3.] and s0.tolist() == [1., 1.]
    s, z, s0, z0 = get_quant_para(ck, 8, "l", "act", split=3)
    assert float(z0) == 0.0
    with pytest.raises(AssertionError):
        get_quant_para(ck, 8, "missing", "act")
    x = torch.tensor([[0.0, 1.0, 2.0, 15.0], [7.0, 8.0, 3.0, 20.0]])
    packed = quantize_per_tensor_uint4(x, torch.tensor([1.0, 1.0]), torch.tensor([0.0, 0.0]))
    assert packed.tolist() == [[0x01, 0x2F], [0x78, 0x3F]]       # high nibble = even index
    assert unpack_uint4(packed).tolist() == [[0, 1, 2, 15], [7, 8, 3, 15]]
    assert dequantize_per_tensor_uint4(packed, torch.tensor([2.0, 1.0]), torch.tensor([1.0, 0.0])
                                       )[0].tolist() == [-2.0, 0.0, 2.0, 28.0]


# ---------------------------------------------------------------------------- tiny UNet swaps
TINY = dict(block_out_channels=(32, 64, 128), transformer_layers_per_block=(0, 1, 2),
            mid_transformer_layers=1, head_dim=16, cross_attention_dim=64, time_embed_dim=128,
            addition_time_embed_dim=16, projection_class_embeddings_input_dim=96 + 32,
            norm_num_groups=8)


def tiny_unet():
    from mixdq_amd.unet import SDXLUNet, init_synthetic_weights
    return init_synthetic_weights(SDXLUNet(TINY)).eval()


def tiny_inputs(B=1, L=8):
    g = torch.Generator().manual_seed(0)
    return dict(sample=torch.rand(B, 4, L, L, generator=g), timestep=torch.tensor(999.),
                encoder_hidden_states=torch.rand(B, 77, 64, generator=g),
                added_cond_kwargs=dict(time_ids=torch.tensor([[64., 64, 0, 0, 64, 64]] * B),
                                       text_embeds=torch.rand(B, 32, generator=g)))


def test_unet_inventory_matches_packed_configs():
    from mixdq_amd import cfgs
    names = cfgs.layer_names()   # asserts the sha of the 794 names
    assert len(names) == 794
    w8, a8 = cfgs.load("weight/uniform_8"), cfgs.load("act/act_8.00")
    assert len(w8) == 794 and set(w8.values()) == {8}
    assert len(a8) == 785
    assert sorted(set(w8) - set(a8)) == [
        "conv_in", "conv_out", "down_blocks.0.resnets.0.conv2",
        "down_blocks.2.attentions.1.transformer_blocks.5.ff.net.2",
        "down_blocks.2.attentions.1.transformer_blocks.6.ff.net.2",
        "down_blocks.2.attentions.1.transformer_blocks.7.ff.net.2",
        "down_blocks.2.attentions.1.transformer_blocks.8.ff.net.2",
        "up_blocks.0.attentions.0.transformer_blocks.0.ff.net.2",
        "up_blocks.2.resnets.2.conv_shortcut"]
    from collections import Counter
    assert Counter(cfgs.load("weight/weight_4.00").values()) == {4: 347, 8: 238, 2: 209}
    assert Counter(cfgs.load("weight/weight_8.00").values()) == {8: 790, 4: 4}
    assert len(cfgs.bos_shapes()) == 140
    import torch.nn as nn_
    from mixdq_amd.unet import SDXLUNet, quantizable_layers
    with torch.device("meta"):
        q = quantizable_layers(SDXLUNet())
    assert sum(isinstance(m, nn_.Linear) for m in q.values()) == 743
    assert sum(isinstance(m, nn_.Conv2d) for m in q.values()) == 51
    splits = [m.split for n, m in q.items() if "up_blocks" in n and "conv_shortcut" in n]
    from mixdq_amd.quantize import SDXL_UP_SHORTCUT_SPLITS
    assert tuple(splits) == SDXL_UP_SHORTCUT_SPLITS          # quantize.py:61 _SPLIT


class Args:
    def __init__(self, w, a):
        self.w_config, self.a_config = w, a


def tiny_quantized(oracle_ops=None, a_drop=()):
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import quantizable_layers
    unet = tiny_unet()
    inp = tiny_inputs()
    with torch.no_grad():
        ref_out = unet(**inp)[0]
    ckpt = calibrate(unet, [inp])
    bos = precompute_bos(unet.half(), inp["encoder_hidden_states"].half())
    names = list(quantizable_layers(unet))
    w_cfg = {"model." + n: 8 for n in names}
    a_cfg = {"model." + n: 8 for n in names if n not in a_drop}
    quantize_unet(unet, Args(w_cfg, a_cfg), ckpt, bos=True, bos_dict=bos)
    return unet, inp, ref_out, ckpt


def test_quantize_unet_swaps_every_layer_and_is_repeatable():
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    for _ in range(2):   # the reference raises IndexError on a second convert (global _NUM)
        unet, inp, ref_out, ckpt = tiny_quantized(a_drop=("conv_in", "conv_out"))
        kinds = [type(m) for m in unet.modules()]
        assert nn.Linear not in kinds and nn.Conv2d not in kinds
        q = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
        assert len(q) == sum(k.endswith(".act_quantizer") for k in ckpt)
        assert not unet.conv_in.valid_for_acceleration        # no a_bit => FP fallback
        assert unet.down_blocks[1].resnets[0].conv1.valid_for_acceleration
        sc = unet.up_blocks[0].resnets[0].conv_shortcut
        assert sc.split == 128 and hasattr(sc, "weight_int_0") and hasattr(sc, "act_scales_0")
        k = unet.down_blocks[1].attentions[0].transformer_blocks[0].attn2.to_k
        assert k.bos is True and k.bos_pre_computed.shape == (1, 1, 64)
        assert not any(hasattr(m, "qconfig") for m in unet.modules())


def test_register_qconfig_rejects_unknown_keys():
    from mixdq_amd.quantize_sdxl import register_qconfig_from_input_files
    unet = tiny_unet()
    with pytest.raises(RuntimeError, match="weight yaml"):
        register_qconfig_from_input_files(unet, Args({"model.nope.layer": 8}, None), False, {})
    unet = tiny_unet()
    with pytest.raises(RuntimeError, match="act yaml"):
        register_qconfig_from_input_files(unet, Args({"model.conv_in": 8}, {"model.nope": 8}),
                                          False, {})


def test_quantized_tiny_unet_tracks_fp_on_cpu(oracle_ops):
    """End to end on CPU with oracle-backed ops (test-only): the W8A8 tiny UNet stays close to
    the FP32 one, i.e. scales, zero points, BOS and split plumbing are consistent."""
    unet, inp, ref_out, _ = tiny_quantized()
    half_inp = dict(sample=inp["sample"].half(), timestep=inp["timestep"],
                    encoder_hidden_states=inp["encoder_hidden_states"].half(),
                    added_cond_kwargs={k: v.half() for k, v in inp["added_cond_kwargs"].items()})
    unet = unet.half()
    with torch.no_grad():
        out = unet(**half_inp)[0].float()
    err = (out - ref_out).abs().max().item()
    assert err < 0.05 * ref_out.abs().max().item() + 0.02, err


def test_calibration_checkpoint_schema():
    from mixdq_amd.calib import calibrate
    unet = tiny_unet()
    ckpt = calibrate(unet, [tiny_inputs(), tiny_inputs(B=2)])
    e = ckpt["conv_in.weight_quantizer"]
    assert e["delta_list"].shape == (3, 32) and e["delta_list"].dtype == torch.float16
    assert torch.all(e["zero_point_list"] == 0)
    a = ckpt["conv_in.act_quantizer"]
    assert a["delta_list"].shape == (3,) and a["zero_point_list"].dtype == torch.float16
    assert 0 <= float(a["zero_point_list"][2]) <= 255
    assert "up_blocks.0.resnets.0.conv_shortcut.act_quantizer_0" in ckpt
    assert "up_blocks.0.resnets.0.conv_shortcut.weight_quantizer_0" in ckpt
    assert ckpt["up_blocks.0.resnets.0.conv_shortcut.weight_quantizer"]["delta_list"].shape == (3, 128)
    # 8-bit step = absmax / 127 of each output channel
    w = unet.conv_in.weight.detach().reshape(32, -1)
    assert torch.equal(e["delta_list"][2], (w.abs().max(dim=1)[0] / 127).half())


def test_convert_ckpt_matches_reference_semantics():
    """kernels/convert_ckpt.py:22-40: keep delta/zero-point lists, drop attention q/k/v activation
    quantizers, fp16, weight -> [3, OC], act -> [3]."""
    from collections import OrderedDict
    from mixdq_amd.convert_ckpt import convert
    buf_w = OrderedDict(delta_list=torch.rand(3, 8, 1, 1, 1), zero_point_list=torch.zeros(3, 8, 1, 1, 1),
                        delta=torch.rand(8, 1, 1, 1), zero_point=torch.zeros(8, 1, 1, 1), alpha=None)
    buf_a = OrderedDict(delta_list=torch.rand(3, 1, 1, 1), zero_point_list=torch.tensor(
        [1., 7., 130.]).reshape(3, 1, 1, 1), delta=torch.rand(1), zero_point=torch.zeros(1))
    ck = OrderedDict([("l.weight_quantizer", [buf_w, OrderedDict()]),
                      ("l.act_quantizer", [buf_a, OrderedDict()]),
                      ("blk.attn1.act_quantizer_q", [buf_a, OrderedDict()]),
                      ("blk.attn1.act_quantizer_k", [buf_a, OrderedDict()]),
                      ("blk.attn1.act_quantizer_v", [buf_a, OrderedDict()])])
    new = convert(ck)
    assert list(new) == ["l.weight_quantizer", "l.act_quantizer"]
    assert new["l.weight_quantizer"]["delta_list"].shape == (3, 8)
    assert new["l.weight_quantizer"]["delta_list"].dtype == torch.float16
    assert new["l.act_quantizer"]["zero_point_list"].tolist() == [1., 7., 130.]
    assert set(new["l.act_quantizer"]) == {"delta_list", "zero_point_list"}
    from mixdq_amd.nn.utils import get_quant_para
    s, z, _, _ = get_quant_para(new, 8, "l", "act")
    assert float(z) == 2.0


# ----------------------------------------------------------------------------- W4A8 (f-2)
def test_w4_pack_layout_and_default_fallback(modules_golden, oracle):
    from mixdq_amd.nn import QuantizedLinear
    from mixdq_amd.nn.utils import pack_w4, unpack_w4
    q = torch.tensor([[-8, -1, 0, 7, 1, 2, 3, -4] * 4], dtype=torch.int8)
    p = pack_w4(q)
    assert p.shape == (1, 16)
    # byte j of a group: high nibble k[j], low nibble k[4+j], two's complement
    assert p.view(torch.uint8)[0, :4].tolist() == [0x81, 0xF2, 0x03, 0x7C]
    assert torch.equal(unpack_w4(p), q)
    assert np.array_equal(oracle.unpack_w4(p.numpy()), q.numpy())
    c = MODULE_CASES[0]
    ck = module_ckpt(c, modules_golden)
    # default: 4-bit weights fall back to FP16 exactly as the reference (nn/Linear.py:31)
    assert not QuantizedLinear.from_float(prepared(c, modules_golden, w_bit=4), ckpt=ck
                                          ).valid_for_acceleration
    fm = prepared(c, modules_golden, w_bit=4)
    fm.w4_kernel = True
    qm = QuantizedLinear.from_float(fm, ckpt=ck)
    assert qm.valid_for_acceleration and qm.w_packed4 and qm._get_name() == "QuantizedLinearW4A8"
    assert "weight_int4" in dict(qm.named_buffers()) and "weight_int" not in dict(qm.named_buffers())
    assert qm.weight_int4.shape == (c["cout"], c["cin"] // 2)
    # the stored integers are the Path A integers: clamp(round(w / delta_4bit), -8, 7)
    w = make_float_module(c).half().weight.detach().float()
    d4 = ck[c["name"] + ".weight_quantizer"]["delta_list"][1].float()
    want = torch.clamp(torch.round(w / d4[:, None]), -8, 7).to(torch.int8)
    assert torch.equal(unpack_w4(qm.weight_int4), want)
    assert torch.equal(qm.bias0, want.float().sum(dim=1) * qm.act_zero_points)


@pytest.mark.parametrize("c", [c for c in MODULE_CASES if c["cin"] % 32 == 0 and not c.get("split")],
                         ids=[c["key"] for c in MODULE_CASES if c["cin"] % 32 == 0 and not c.get("split")])
def test_w4_forward_tracks_path_a_4bit(modules_golden, fakequant_golden, oracle_ops, c):
    """W4A8 module (oracle-backed ops on CPU) vs the reference QuantLayer at 4-bit weights."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
    fm = prepared(c, modules_golden, w_bit=4)
    fm.w4_kernel = True
    qm = cls.from_float(fm, ckpt=module_ckpt(c, modules_golden))
    assert qm.valid_for_acceleration and qm.w_packed4
    with torch.no_grad():
        y = qm(module_input(c)).float()
    # Path A restated (oracle/fakequant.py, pinned to the reference QuantLayer) with the SAME fp16
    # checkpoint scales the module uses: rtol = atol = 1e-2 (the reference's tolerance)
    from oracle.fakequant import quant_layer_forward
    key = c["key"]

    def ck(sfx, field, i):
        return torch.from_numpy(modules_golden[f"{key}.ckpt.{sfx}.{field}"]).float()[i]

    fm32 = make_float_module(c).half().float()
    kw = None
    if c["kind"] == "conv":
        kw = dict(stride=fm32.stride, padding=fm32.padding, dilation=fm32.dilation, groups=1)
    with torch.no_grad():
        sim = quant_layer_forward(module_input(c).float(), fm32.weight, fm32.bias,
                                  ck("weight_quantizer", "delta_list", 1),
                                  ck("act_quantizer", "delta_list", 2),
                                  ck("act_quantizer", "zero_point_list", 2), 4, 8, kw)
    ref = torch.from_numpy(fakequant_golden[f"{key}.pathA_w4a8"])
    if c.get("bos"):
        y, sim, ref = y[:, 1:], sim[:, 1:], ref[:, 1:]
    torch.testing.assert_close(y, sim, rtol=1e-2, atol=1e-2)
    # vs the reference QuantLayer's own output (fp32 scales; the fp16 rounding of a 4-bit step can
    # move a weight by one LSB = delta, hence the wider bound on isolated elements)
    assert (y - ref).abs().mean().item() <= 2e-3 * ref.abs().max().item() + 1e-3


def test_w2_layers_clamp_to_their_own_range_in_4bit_storage(modules_golden):
    """weight_4.00.yaml has 209 two-bit layers: Path A clamps them to [-2, 1] with the 2-bit delta
    (base_quantizer.py:119-129); the packed-W4 storage must hold exactly those integers."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from mixdq_amd.nn.utils import unpack_w4
    for c in (MODULE_CASES[0], next(c for c in MODULE_CASES if c["kind"] == "conv"
                                    and c["cin"] % 32 == 0 and not c.get("split"))):
        cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
        ck = module_ckpt(c, modules_golden)
        fm = prepared(c, modules_golden, w_bit=2)
        fm.w4_kernel = True
        qm = cls.from_float(fm, ckpt=ck)
        assert qm.valid_for_acceleration and qm.w_packed4
        w = make_float_module(c).half().weight.detach().float()
        d2 = ck[c["name"] + ".weight_quantizer"]["delta_list"][0].float()
        d2 = d2.reshape(-1, *([1] * (w.dim() - 1)))
        want = torch.clamp(torch.round(w / d2), -2, 1).to(torch.int8)
        got = qm._weight_values()
        assert torch.equal(got, want)
        assert int(got.min()) >= -2 and int(got.max()) <= 1


# ----------------------------------------------------------------------------- fused-graph caches
def test_packed_qkv_is_storage_not_a_copy_and_is_retaken_after_a_module_swap():
    """(1) nothing negative is cached -- a fused FP16 forward before quantize_unet must
    not pin the unfused launches; (2) the packed operands are the storage (layers keep views), so
    load_state_dict / broadcast reach the packed GEMM; (3) swapping a layer invalidates the pack."""
    import mixdq_amd.unet as U
    from mixdq_amd.nn import QuantizedLinear
    unet, inp, _, _ = tiny_quantized()
    blk = next(m for m in unet.modules() if isinstance(m, U.BasicTransformerBlock))
    pack = blk._qkv_fused()
    assert pack is not None and blk._qkv_fused() is pack
    a = blk.attn1
    C = a.to_q.out_features
    for i, m in enumerate((a.to_q, a.to_k, a.to_v)):
        for key, name in pack["names"]:
            t = getattr(m, name)
            assert t.data_ptr() == pack[key][i * C:(i + 1) * C].data_ptr()
            assert name in dict(m.named_buffers())          # still a registered buffer
    # (2) in-place update of a layer's buffer is seen through the pack
    with torch.no_grad():
        a.to_k.scale.mul_(2.0)
        sd = unet.state_dict()
        sd = {k: (v * 0 + 3 if k.endswith("attn1.to_v.bias0") and "down_blocks.1.attentions.0."
                  "transformer_blocks.0." in k else v) for k, v in sd.items()}
        unet.load_state_dict(sd)
    assert torch.equal(pack["scale"][C:2 * C], a.to_k.scale)
    assert bool((pack["bias0"][2 * C:] == 3).all())
    assert blk._qkv_fused() is pack
    # (3) a swapped layer (new module object, new tensors) => a new pack
    fresh = QuantizedLinear.__new__(QuantizedLinear)
    nn.Module.__init__(fresh)
    fresh.__dict__.update({k: v for k, v in a.to_q.__dict__.items() if not k.startswith("_")})
    for n, b in a.to_q.named_buffers():
        fresh.register_buffer(n, b.clone())
    fresh.bias = None
    a.to_q = fresh
    pack2 = blk._qkv_fused()
    assert pack2 is not None and pack2 is not pack
    assert a.to_q.scale.data_ptr() == pack2["scale"].data_ptr()
    # different activation quantizer => not packable, and that answer is not sticky
    with torch.no_grad():
        a.to_v.act_zero_points.add_(1.0)
    assert blk._qkv_fused() is None
    with torch.no_grad():
        a.to_v.act_zero_points.sub_(1.0)
    assert blk._qkv_fused() is not None


def test_mixed_w4_w8_groups_are_unified_by_set_fused_never_by_a_forward():
    """widening a packed 4-bit layer to int8 storage changes state-dict keys; it is an
    explicit step of set_fused (prepare_fused_), not a side effect of the first fused forward."""
    import mixdq_amd.unet as U
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import quantize_unet
    from mixdq_amd.unet import quantizable_layers
    unet = tiny_unet()
    inp = tiny_inputs()
    ckpt = calibrate(unet, [inp])
    bos = precompute_bos(unet.half(), inp["encoder_hidden_states"].half())
    names = list(quantizable_layers(unet))
    mixed = lambda n: 4 if (n.endswith("attn1.to_k") or n.endswith("attn2.to_v")) else 8   # noqa: E731
    quantize_unet(unet, Args({"model." + n: mixed(n) for n in names},
                             {"model." + n: 8 for n in names}), ckpt, bos=True, bos_dict=bos,
                  w4_kernel=True)
    blk = next(m for m in unet.modules() if isinstance(m, U.BasicTransformerBlock))
    assert blk.attn1.to_k.w_packed4 and not blk.attn1.to_q.w_packed4
    keys0 = sorted(unet.state_dict())
    assert any(k.endswith("attn1.to_k.weight_int4") for k in keys0)
    # the forward-side queries are pure: a mixed group is simply not packed
    assert blk._qkv_fused() is None and U.SDXLUNet._kv_pack(blk) is None
    assert sorted(unet.state_dict()) == keys0
    values = blk.attn1.to_k._weight_values().clone()
    unet.set_fused(True)                        # the explicit step
    keys1 = sorted(unet.state_dict())
    assert keys1 != keys0 and not any(k.endswith("to_k.weight_int4") or k.endswith("to_v.weight_int4")
                                      for k in keys1 if "attn1.to_k" in k or "attn2.to_v" in k)
    assert not blk.attn1.to_k.w_packed4 and torch.equal(blk.attn1.to_k.weight_int, values)
    assert blk._qkv_fused() is not None and U.SDXLUNet._kv_pack(blk) is not None
    assert sorted(unet.state_dict()) == keys1   # ... and stable from here on
    unet.set_fused(True)
    assert sorted(unet.state_dict()) == keys1


def test_fused_flags_on_plain_float_network_do_not_stick():
    import mixdq_amd.unet as U
    unet = tiny_unet()
    unet.set_fused(True)
    blk = next(m for m in unet.modules() if isinstance(m, U.BasicTransformerBlock))
    assert blk._qkv_fused() is None and U.SDXLUNet._kv_pack(blk) is None   # nn.Linear: no pack
    unet.set_fused(False)
    unet2, _, _, _ = tiny_quantized()
    blk2 = next(m for m in unet2.modules() if isinstance(m, U.BasicTransformerBlock))
    assert blk2._qkv_fused() is not None and U.SDXLUNet._kv_pack(blk2) is not None


def test_quantizer_groups_memo_follows_identity_and_version():
    import mixdq_amd.unet as U

    class L:
        def __init__(self, s, z):
            self.act_scales_inv, self.act_zero_points = torch.tensor([s]), torch.tensor([z])
    a, b, c = L(2.0, 1.0), L(2.0, 1.0), L(2.0, 3.0)
    h = {}
    assert U._quantizer_groups(h, "k", [a, b, c]) == [0, 0, 1]
    first = h["k"][2]
    assert U._quantizer_groups(h, "k", [a, b, c]) is first        # memo hit
    c.act_zero_points.fill_(1.0)                                    # in-place update
    assert U._quantizer_groups(h, "k", [a, b, c]) == [0, 0, 0]
    b.act_scales_inv = torch.tensor([4.0])                          # replaced tensor
    assert U._quantizer_groups(h, "k", [a, b, c]) == [0, 1, 0]
    assert U._quantizer_groups(h, "k", []) == []


def test_geglu_row_order_is_value_gate_groups_of_16():
    """include/mixdq_hip.h, mixdq_qlinear_w8a8_geglu: stored row i = row perm[i] of the ordinary
    [values 0..D-1 ; gates D..2D-1] projection, in groups [v 0..15 | g 0..15 | v 16..31 | g 16..31 ...] --
    every 32 stored rows hold the value and the gate of 16 outputs (what lets one lane of an MFMA tile hold
    both halves of an output)."""
    from mixdq_amd._C import geglu_row_order
    D = 80
    perm = geglu_row_order(D).tolist()
    assert sorted(perm) == list(range(2 * D))
    for grp in range(D // 16):
        blk = perm[32 * grp:32 * grp + 32]
        assert blk[:16] == list(range(16 * grp, 16 * grp + 16))
        assert blk[16:] == list(range(D + 16 * grp, D + 16 * grp + 16))
    inv = torch.argsort(torch.tensor(perm))
    assert torch.equal(torch.tensor(perm)[inv], torch.arange(2 * D))


def test_prefetch_plan_own_interval_first_bounded_look_back_and_row_limit():
    """mixdq_amd.unet._build_prefetch_plan: launch j reads ahead the weights used between
    it and launch j + 1, in order, within its byte budget; what does not fit is offered to at most
    PREFETCH_MAX_LEAD earlier launches behind their own intervals, then dropped; launches over the row limit
    carry nothing; weights before the first launch have no host; small tensors and repeats are skipped."""
    from mixdq_amd import unet as U

    def w(mb):
        return torch.empty(int(mb * 1e6), dtype=torch.int8)
    a = ("attn", 1024, 1024)                                     # budget = PREFETCH_MB_PER_LAUNCH (48 MB)
    early, w1, w2, w3, big, small, tail = w(5), w(20), w(20), w(20), w(60), w(0.01), w(10)
    trace = [early, a, w1, small, a, w2, w3, w1, a, big, tail]
    plan = U._build_prefetch_plan(trace)
    ptrs = [[ref().data_ptr() for ref in lst] for lst in plan["lists"]]      # the plan holds weak references
    assert ptrs[2] == []                                         # 60 MB does not fit; `tail` may not overtake it
    assert ptrs[1] == [w2.data_ptr(), w3.data_ptr()]             # own interval (w1 repeated: once), 40 of 48 MB; ...
    assert ptrs[0] == [w1.data_ptr(), tail.data_ptr()]           # ... `tail` (10 MB) moves back behind w1; big is dropped
    assert all(early.data_ptr() not in p and small.data_ptr() not in p and big.data_ptr() not in p for p in ptrs)
    assert plan["sig"] == U._trace_signature(trace) and U._plan_alive(plan, torch.device("cpu"))
    saved, U.PREFETCH_MAX_ROWS = U.PREFETCH_MAX_ROWS, 4096                   # the optional row limit
    try:
        over = ("attn", 8192, 1024)
        assert U._build_prefetch_plan([over, w1, over, w2])["lists"] == [[], []]
    finally:
        U.PREFETCH_MAX_ROWS = saved
    assert U._build_prefetch_plan([w1, w2]) is None              # no attention launch: no plan
    lead = U.PREFETCH_MAX_LEAD
    chain = []
    for _ in range(lead + 2):
        chain += [a, w(40)]                                      # every launch is full with its own 40 MB ...
    last = w(30)
    plan = U._build_prefetch_plan(chain + [last])                # ... so a 30 MB tensor behind the last finds no room
    assert all(last.data_ptr() not in [ref().data_ptr() for ref in lst] for lst in plan["lists"])
    # a plan never keeps a replaced weight alive, and notices that it is gone
    keep = [w(30), w(30)]
    plan = U._build_prefetch_plan([a, keep[0], a, keep[1]])
    assert U._plan_alive(plan, torch.device("cpu"))
    del keep[1]
    import gc
    gc.collect()
    assert not U._plan_alive(plan, torch.device("cpu"))


def test_prefetch_state_is_per_forward_and_per_thread():
    """mixdq_amd._C.PrefetchContext: the weight trace / plan / launch counter of a forward live in
    an object that is active on ONE thread while that forward runs -- two UNets whose forwards interleave
    (threads, one per device) cannot mix their traces; a nested forward leaves the outer context in charge;
    tensors of another device are neither traced nor handed to a launch."""
    import threading
    import weakref
    from mixdq_amd import _C
    assert _C.prefetch_context() is None and not hasattr(_C, "TRACE") and not hasattr(_C, "PLAN")
    wa, wb = torch.empty(4, dtype=torch.int8), torch.empty(8, dtype=torch.int8)
    ctx_a = _C.PrefetchContext("cpu", plan=[[weakref.ref(wa)], [weakref.ref(wb)]])
    with ctx_a:
        assert _C.prefetch_context() is ctx_a
        _C._trace_w(wa)
        inner = _C.PrefetchContext("cpu")
        with inner:                                   # nested: a no-op, the outer one keeps tracing
            assert _C.prefetch_context() is ctx_a
            _C._trace_w(wb)
        assert _C.prefetch_context() is ctx_a and inner.trace == []
        seen = {}

        def other_thread():                           # another thread sees no context, and gets its own
            seen["before"] = _C.prefetch_context()
            with _C.PrefetchContext("cpu") as c:
                _C._trace_w(wb)
                seen["own"] = c.trace
            seen["after"] = _C.prefetch_context()
        th = threading.Thread(target=other_thread)
        th.start()
        th.join()
        assert seen["before"] is None and seen["after"] is None and len(seen["own"]) == 1
        assert [t.data_ptr() for t in ctx_a.trace] == [wa.data_ptr(), wb.data_ptr()]
        assert [t.data_ptr() for t in ctx_a.payload()] == [wa.data_ptr()]      # launch 0, then launch 1 ...
        assert [t.data_ptr() for t in ctx_a.payload()] == [wb.data_ptr()]
        assert ctx_a.payload() is None                                            # ... then nothing planned
    assert _C.prefetch_context() is None
    # another device's tensors: not traced, never in a payload
    ctx_m = _C.PrefetchContext("meta", plan=[[weakref.ref(wa)]])
    with ctx_m:
        _C._trace_w(wa)
        assert ctx_m.trace == [] and ctx_m.payload() == []


def test_f16in_rows_view_accepts_exactly_the_operands_the_kernel_can_read_in_place():
    """mixdq_amd._C._rows_view (host logic of the quantize-in-prologue GEMM): an FP16 operand is read in place when
    its rows of K contiguous values are a CONSTANT stride apart -- a dense tensor with any leading shape, a
    last-dimension slice of one (the split shortcut's halves over NHWC rows) -- and nothing else: the BOS slice
    x[:, 1:, :] at batch > 1 (two strides) goes through the row map, a transposed operand through two launches."""
    from mixdq_amd._C import _rows_view
    x = torch.zeros(2, 77, 64, dtype=torch.float16)
    assert _rows_view(x, 64) == (154, 64, [2, 77])
    assert _rows_view(x[..., :32], 32) == (154, 64, [2, 77])                  # column slice: lda = all columns
    assert _rows_view(x[..., 32:], 32) == (154, 64, [2, 77])
    assert _rows_view(x[:1, 1:, :], 64) == (76, 64, [1, 76])                  # batch 1: one stride
    assert _rows_view(x[:, 1:, :], 64) is None                                # batch 2: rows are not equidistant
    assert _rows_view(x.transpose(1, 2), 77) is None                          # last dimension not contiguous
    assert _rows_view(x, 32) is None                                          # K must be the last dimension
    nhwc = torch.zeros(2, 96, 6, 6, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    rows = nhwc[:, :64].permute(0, 2, 3, 1)                                   # the first half of a split shortcut
    assert _rows_view(rows, 64) == (72, 96, [2, 6, 6])
    assert _rows_view(torch.zeros(0, 64, dtype=torch.float16), 64) is None
    assert _rows_view(torch.zeros(1, 64, dtype=torch.float16), 64) == (1, 64, [1])


def test_swap_glue_modules_by_type_keeps_names_and_state_and_undoes():
    """quantize_unet(..., swap_glue=True) / mixdq_amd.nn.glue (host logic; the kernels are checked on the GPU,
    tests/test_glue_gpu.py): module CLASSES change to subclasses of the stock ones, names / state_dict keys /
    isinstance do not; the SiLU behind a ResnetBlock2D's GroupNorms and behind conv_norm_out is folded, the one in
    front of time_emb_proj keeps acting; idempotent; undone by unswap_glue_modules; on tensors the kernels do not take
    (here: the CPU, FP32) every swapped module IS its stock op."""
    import bench
    import torch.nn as nn
    from mixdq_amd.nn.glue import (HipGroupNorm, HipLayerNorm, HipSiLU, swap_glue_modules, unswap_glue_modules)
    from mixdq_amd.quantize_sdxl import example_inputs
    from mixdq_amd.unet import GEGLU, Attention, SDXLUNet, init_synthetic_weights
    unet = init_synthetic_weights(SDXLUNet(bench.TINY_CFG)).eval()
    inputs = example_inputs(1, 8, "cpu", seed=3)
    inputs = {k: (v.float() if torch.is_tensor(v) else {a: b.float() for a, b in v.items()}) for k, v in inputs.items()}
    with torch.no_grad():
        want = unet(**inputs)[0]
    keys, names = list(unet.state_dict()), [n for n, _ in unet.named_modules()]
    n = swap_glue_modules(unet)
    n_res = sum(1 for m in unet.modules() if type(m).__name__ == "ResnetBlock2D")
    assert n["groupnorm"] == sum(isinstance(m, nn.GroupNorm) for m in unet.modules())
    assert n["silu_folded"] == 2 * n_res + 1 and n["layernorm"] > 0 and n["geglu"] > 0 and n["attention"] > 0
    assert list(unet.state_dict()) == keys and [m for m, _ in unet.named_modules()] == names
    res = next(m for m in unet.modules() if type(m).__name__ == "ResnetBlock2D")
    assert type(res.norm1) is HipGroupNorm and res.norm1.fuse_silu and type(res.nonlinearity) is HipSiLU
    assert type(unet.conv_norm_out) is HipGroupNorm and unet.conv_norm_out.fuse_silu and type(unet.conv_act) is HipSiLU
    tnorm = next(m for nme, m in unet.named_modules() if nme.endswith("attentions.0.norm"))
    assert type(tnorm) is HipGroupNorm and not tnorm.fuse_silu
    assert all(isinstance(m, (nn.LayerNorm, HipLayerNorm)) for m in unet.modules() if "LayerNorm" in type(m).__name__)
    assert all(isinstance(m, GEGLU) for m in unet.modules() if "GEGLU" in type(m).__name__)
    assert all(isinstance(m, Attention) for m in unet.modules() if "Attention" in type(m).__name__)
    # operand hand-off links (OPERAND_PAIRS): per transformer block norm1 -> q / k / v, norm2 -> to_q, norm3 -> the GEGLU
    # projection, GEGLU -> net.2; per ResNet block norm1 -> conv1, norm2 -> conv2; plain references, not sub-modules
    n_tb = sum(1 for m in unet.modules() if type(m).__name__ == "BasicTransformerBlock")
    assert n["operand_links"] == 6 * n_tb + 2 * n_res and n["attention_handoff"] == n["attention"] == 2 * n_tb
    tb = next(m for m in unet.modules() if type(m).__name__ == "BasicTransformerBlock")
    assert tb.norm1.__dict__["_mixdq_consumers"] == (tb.attn1.to_q, tb.attn1.to_k, tb.attn1.to_v)
    assert tb.ff.net[0].__dict__["_mixdq_consumers"] == (tb.ff.net[2],) and tb.attn1.hand_off
    assert res.norm2.__dict__["_mixdq_consumers"] == (res.conv2,) and "_mixdq_consumers" not in tnorm.__dict__
    assert swap_glue_modules(unet) == dict(groupnorm=0, silu_folded=0, layernorm=0, geglu=0, attention=0,
                                           operand_links=0, attention_handoff=0)
    with torch.no_grad():
        assert torch.equal(unet(**inputs)[0], want)          # CPU / FP32: every swapped module runs its stock op
    unswap_glue_modules(unet)
    assert {type(m) for m in unet.modules() if isinstance(m, (nn.GroupNorm, nn.LayerNorm, nn.SiLU))} == \
        {nn.GroupNorm, nn.LayerNorm, nn.SiLU}
    assert not any("Hip" in type(m).__name__ for m in unet.modules())
    assert not any("_mixdq_consumers" in m.__dict__ or "hand_off" in m.__dict__ for m in unet.modules())
    n2 = swap_glue_modules(unet, attention=False, operands=False)
    assert n2["attention"] == 0 and n2["groupnorm"] == n["groupnorm"] and n2["operand_links"] == 0
    assert not any("_mixdq_consumers" in m.__dict__ for m in unet.modules())
    n3 = swap_glue_modules(unet, attention=False)               # the links can be added to an earlier swap
    assert n3["operand_links"] == n["operand_links"] and n3["groupnorm"] == 0 and n3["attention_handoff"] == 0
    assert not any(type(m).__name__ == "HipAttention" for m in unet.modules())


def test_pack_static_moves_every_tensor_into_one_allocation_and_keeps_the_network():
    """mixdq_amd/arena.py: same tensor objects, names, values, aliasing (row packs) -- one storage."""
    import bench
    from mixdq_amd.arena import ALIGN, pack_static_
    from mixdq_amd.quantize_sdxl import example_inputs
    from mixdq_amd.unet import SDXLUNet, init_synthetic_weights
    unet = init_synthetic_weights(SDXLUNet(bench.TINY_CFG)).eval()
    inputs = example_inputs(1, 8, "cpu", seed=5)
    inputs = {k: (v.float() if torch.is_tensor(v) else {a: b.float() for a, b in v.items()}) for k, v in inputs.items()}
    lin = next(m for m in unet.modules() if isinstance(m, nn.Linear))
    cat = torch.arange(24, dtype=torch.float32)
    lin.register_buffer("lo", cat[:8])                       # two buffers and a row pack on one storage
    lin.register_buffer("hi", cat[8:].view(2, 8).t())
    lin.__dict__["_qkv"] = dict(layers=(), names=(), w=cat)
    with torch.no_grad():
        want = unet(**inputs)[0]
    state = {k: v.clone() for k, v in unet.state_dict().items()}
    ids = {k: id(v) for k, v in list(unet.named_parameters()) + list(unet.named_buffers())}
    r = pack_static_(unet)
    tensors = list(unet.parameters()) + list(unet.buffers()) + [cat]
    assert r["tensors"] == len(tensors) and r["storages"] == len(tensors) - 2
    assert len({t.untyped_storage().data_ptr() for t in tensors}) == 1                # one allocation
    assert r["bytes"] == tensors[0].untyped_storage().nbytes() and r["bytes"] % ALIGN == 0
    assert lin.lo.data_ptr() == cat.data_ptr() and lin.hi.data_ptr() == cat.data_ptr() + 32
    assert lin.hi.stride() == (1, 8) and torch.equal(cat, torch.arange(24, dtype=torch.float32))
    assert {k: id(v) for k, v in list(unet.named_parameters()) + list(unet.named_buffers())} == ids
    assert all(isinstance(p, nn.Parameter) for p in unet.parameters())
    for k, v in unet.state_dict().items():
        assert torch.equal(v, state[k]), k
    spans = sorted((t.data_ptr(), t.untyped_storage().data_ptr()) for t in tensors)
    assert all((a - base) % 4 == 0 for a, base in spans)
    with torch.no_grad():
        assert torch.equal(unet(**inputs)[0], want)
    assert pack_static_(nn.Identity()) == dict(bytes=0, storages=0, tensors=0)


def test_tagged_operand_matches_tensor_object_quantizer_buffers_and_version():
    """nn/glue.py tagged_operand / _attach on the CPU: the hand-off is found only on the tensor it was attached to,
    unmodified since, for the layer whose own quantizer buffers it was made with."""
    from mixdq_amd.nn.glue import _attach, tagged_operand

    class L:
        def __init__(self):
            self.act_scales_inv, self.act_zero_points = torch.tensor(20.0), torch.tensor(3.0)
    a, b = L(), L()
    y, q = torch.zeros(2, 4, dtype=torch.float16), torch.zeros(2, 4, dtype=torch.int8)
    assert _attach(y, [a, b], [q, None]) is y
    assert tagged_operand(y, a) is q and tagged_operand(y, b) is None
    assert tagged_operand(y.clone(), a) is None and tagged_operand(torch.zeros(2, 4), a) is None
    a2 = L()
    a2.act_scales_inv = a.act_scales_inv                      # one of the two buffers only
    assert tagged_operand(y, a2) is None
    y.mul_(1)
    assert tagged_operand(y, a) is None
    z = torch.zeros(2, 4, dtype=torch.float16)
    assert not hasattr(_attach(z, [a], [None]), "_mixdq_operands")
    _attach(z, [a], [torch.zeros(8, dtype=torch.int8)])       # a shape that is not the tensor's: never used
    assert tagged_operand(z, a) is None


def test_kept_bos_buffers_are_bounded_in_number_and_refilled_when_the_row_changes(monkeypatch):
    """nn/glue.py _bos_buffer (host logic; the GPU side: tests/test_modules_gpu.py): one buffer per (batch, tokens,
    device), never replaced; past BOS_BUFFERS_MAX shapes None (the caller then runs the reference's allocate + copy);
    row 0 re-filled in place when the BOS row was replaced or modified."""
    import mixdq_amd.nn.glue as G

    class L(nn.Module):
        def __init__(self):
            super().__init__()
            self.out_features = 8
            self.register_buffer("bos_pre_computed", torch.arange(8, dtype=torch.float16).view(1, 1, 8))
    m = L()
    monkeypatch.setattr(G, "BOS_BUFFERS_MAX", 2)
    a = G._bos_buffer(m, torch.zeros(2, 5, 4, dtype=torch.float16))
    assert a.shape == (2, 5, 8) and torch.equal(a[:, 0], m.bos_pre_computed.expand(2, 1, 8)[:, 0])
    assert G._bos_buffer(m, torch.zeros(2, 5, 4, dtype=torch.float16)) is a
    b = G._bos_buffer(m, torch.zeros(3, 5, 4, dtype=torch.float16))
    assert b is not a and b.shape == (3, 5, 8)
    assert G._bos_buffer(m, torch.zeros(4, 5, 4, dtype=torch.float16)) is None       # the third shape: not kept
    assert G._bos_buffer(m, torch.zeros(2, 5, 4, dtype=torch.float16)) is a           # the kept ones still served
    a[:, 1:] = 7
    m.bos_pre_computed.mul_(2)                                                          # modified in place
    a2 = G._bos_buffer(m, torch.zeros(2, 5, 4, dtype=torch.float16))
    assert a2 is a and torch.equal(a[:, 0], m.bos_pre_computed.expand(2, 1, 8)[:, 0]) and (a[:, 1:] == 7).all()
    m.bos_pre_computed = torch.ones(1, 1, 8, dtype=torch.float16)                      # replaced
    assert torch.equal(G._bos_buffer(m, torch.zeros(2, 5, 4, dtype=torch.float16))[:, 0], torch.ones(2, 8, dtype=torch.float16))
    G.unswap_glue_modules(m)
    assert "_mixdq_bos_out" not in m.__dict__
