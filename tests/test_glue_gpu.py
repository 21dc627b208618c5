"""`quantize_unet(..., swap_glue=True)` (mixdq_amd/nn/glue.py): the stock glue modules between the quantized layers --
nn.GroupNorm (+ the nn.SiLU behind it), nn.LayerNorm, GEGLU, the FP16 attention core -- swapped by type for this
repo's FP16-output kernels behind the reference's module-swap surface (quantize_sdxl.py:142-156).  Each swapped module

* is bit-equal to the FP16 output (`out_h`) of the kernel the fused graph runs, which the oracle pins
  (tests/test_fused_gpu.py), and
* is within one FP16 ulp per rounding point of PyTorch's FP32-reference op (the stock module evaluated in FP32);

and the swapped drop-in UNet == the same network with every swapped module replaced by that kernel called directly
(the de-fused reference of the fused graph), bit for bit, while staying within quantization noise of the stock one."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests import detdata as dd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def ulp_f16(ref):
    a = ref.float().abs().clamp(min=2.0 ** -14)
    return 2.0 ** (torch.floor(torch.log2(a)) - 10)


class _Res(nn.Module):
    """The part of a ResnetBlock2D the swap looks at: norm -> nonlinearity, the module pair of diffusers."""

    def __init__(self, c, g):
        super().__init__()
        self.norm1 = nn.GroupNorm(g, c, eps=1e-5)
        self.norm2 = nn.GroupNorm(g, c, eps=1e-5)
        self.nonlinearity = nn.SiLU()

    def forward(self, x, temb):
        return self.nonlinearity(self.norm1(x)), self.nonlinearity(self.norm2(x)), self.nonlinearity(temb)


@pytest.mark.parametrize("N,H,W,Cc,G", [(1, 16, 16, 320, 32), (2, 8, 8, 640, 32), (1, 32, 32, 1280, 32), (3, 5, 7, 64, 8)])
def test_groupnorm_silu_pair_swapped(C, N, H, W, Cc, G):
    from mixdq_amd.nn.glue import HipGroupNorm, HipSiLU, swap_glue_modules, unswap_glue_modules
    m = _Res(Cc, G).half().to(DEV)
    with torch.no_grad():
        m.norm1.weight.copy_(t((dd.normal_f16(51, (Cc,), 0.3).astype(np.float32) + 1).astype(np.float16)))
        m.norm1.bias.copy_(t(dd.normal_f16(52, (Cc,), 0.2)))
        m.norm2.weight.copy_(m.norm1.weight)
        m.norm2.bias.copy_(m.norm1.bias)
    x = t(dd.normal_f16(53, (N, H, W, Cc), 1.5)).permute(0, 3, 1, 2)       # NCHW view of NHWC memory
    temb = t(dd.normal_f16(54, (N, 1280), 1.0))
    keys = list(m.state_dict())
    # (folding is limited to parents KNOWN to apply the SiLU module to the norm's output: here the stand-in is named)
    pairs = {"_Res": (("norm1", "norm2"), "nonlinearity")}
    assert swap_glue_modules(_Res(Cc, G))["silu_folded"] == 0          # an unknown parent class: nothing folded
    n = swap_glue_modules(m, silu_pairs=pairs)
    assert n["groupnorm"] == 2 and n["silu_folded"] == 2 and list(m.state_dict()) == keys
    assert type(m.norm1) is HipGroupNorm and type(m.nonlinearity) is HipSiLU and isinstance(m.norm1, nn.GroupNorm)
    with torch.no_grad():
        a1, a2, at = m(x, temb)
    # == the kernel the fused graph runs (its FP16 output), one launch: GroupNorm + SiLU
    want = C.groupnorm_silu_quantize(x, G, m.norm1.weight, m.norm1.bias, 1e-5, silu=True, want_f16=True)[1]
    assert torch.equal(a1.view(torch.int16), want.view(torch.int16)) and torch.equal(a1, a2)
    assert a1.shape == x.shape and a1.is_contiguous(memory_format=torch.channels_last)
    # the SiLU module still acts on tensors that did not come out of a folded GroupNorm (the time embedding)
    assert torch.equal(at, F.silu(temb))
    # vs PyTorch's FP32-reference ops, one rounding point at a time (as tests/test_fused_gpu.py does for the kernel)
    pre = C.groupnorm_silu_quantize(x, G, m.norm1.weight, m.norm1.bias, 1e-5, silu=False, want_f16=True)[1]
    ref = F.group_norm(x.float(), G, m.norm1.weight.float(), m.norm1.bias.float(), 1e-5).half()
    assert ((pre.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref) + 2e-6).all()
    ref = F.silu(pre.float()).half()
    assert ((a1.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref)).all()
    # an input the kernel does not take (FP32; NCHW memory) goes to the stock op, activation applied once
    with torch.no_grad():
        m32 = _Res(Cc, G).to(DEV)
        swap_glue_modules(m32, silu_pairs=pairs)
        y32 = m32(x.float(), temb.float())[0]
        assert torch.allclose(y32, F.silu(F.group_norm(x.float(), G, m32.norm1.weight, m32.norm1.bias, 1e-5)), atol=1e-6)
        if N * H * W > 1:
            xc = x.contiguous()                                               # NCHW memory
            y = m(xc, temb)[0]
            assert torch.equal(y, F.silu(F.group_norm(xc, G, m.norm1.weight, m.norm1.bias, 1e-5)))
    unswap_glue_modules(m)
    assert type(m.norm1) is nn.GroupNorm and type(m.nonlinearity) is nn.SiLU and "fuse_silu" not in m.norm1.__dict__


def test_groupnorm_without_a_silu_behind_it_is_not_folded(C):
    """Transformer2DModel.norm: a GroupNorm whose parent applies no activation."""
    from mixdq_amd.nn.glue import swap_glue_modules
    class P(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = nn.GroupNorm(32, 640, eps=1e-6)
    m = P().half().to(DEV)
    assert swap_glue_modules(m) == dict(groupnorm=1, silu_folded=0, layernorm=0, geglu=0, attention=0, operand_links=0, attention_handoff=0)
    x = t(dd.normal_f16(55, (2, 8, 8, 640), 1.5)).permute(0, 3, 1, 2)
    with torch.no_grad():
        y = m.norm(x)
    want = C.groupnorm_silu_quantize(x, 32, m.norm.weight, m.norm.bias, 1e-6, silu=False, want_f16=True)[1]
    assert torch.equal(y.view(torch.int16), want.view(torch.int16)) and not getattr(y, "_mixdq_silu_applied", False)


@pytest.mark.parametrize("M,Cc", [(1024, 1280), (4096, 640), (77, 2048), (5, 48)])
def test_layernorm_swapped(C, M, Cc):
    from mixdq_amd.nn.glue import HipLayerNorm, swap_glue_modules
    m = nn.Sequential(nn.LayerNorm(Cc)).half().to(DEV)
    with torch.no_grad():
        m[0].weight.copy_(t((dd.normal_f16(61, (Cc,), 0.3).astype(np.float32) + 1).astype(np.float16)))
        m[0].bias.copy_(t(dd.normal_f16(62, (Cc,), 0.2)))
    assert swap_glue_modules(m)["layernorm"] == 1 and type(m[0]) is HipLayerNorm
    x = t((dd.normal_f16(63, (2, M, Cc), 1.2).astype(np.float32) + 0.4).astype(np.float16))
    with torch.no_grad():
        y = m(x)
    want = C.layernorm_quantize(x, m[0].weight, m[0].bias, 1e-5, [], want_f16=True)[1]
    assert torch.equal(y.view(torch.int16), want.view(torch.int16)) and y.shape == x.shape
    ref = F.layer_norm(x.float(), (Cc,), m[0].weight.float(), m[0].bias.float(), 1e-5).half()
    assert ((y.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref) + 2e-6).all()
    with torch.no_grad():                       # a strided view: the stock op
        xs = x[:, ::2]
        assert torch.equal(m(xs), F.layer_norm(xs, (Cc,), m[0].weight, m[0].bias, 1e-5))


def test_geglu_swapped(C):
    from mixdq_amd.nn.glue import swap_glue_modules
    from mixdq_amd.unet import GEGLU
    m = nn.Sequential(GEGLU(640, 2560)).half().to(DEV)
    keys = list(m.state_dict())
    assert swap_glue_modules(m)["geglu"] == 1 and isinstance(m[0], GEGLU) and list(m.state_dict()) == keys
    x = t(dd.normal_f16(71, (2, 1024, 640), 1.0))
    with torch.no_grad():
        y = m(x)
        h = m[0].proj(x)
    want = C.geglu_quantize(h, want_f16=True)[1]
    assert torch.equal(y.view(torch.int16), want.view(torch.int16))
    D = 2560
    ref = (h[..., :D].float() * F.gelu(h[..., D:].float()).half().float()).half()
    atol = 4e-7 * h[..., :D].float().abs() * h[..., D:].float().abs().clamp(min=1.0)
    assert ((y.float() - ref.float()).abs() <= 2.001 * ulp_f16(ref) + atol).all()


def test_attention_core_swapped_and_diffusers_processor(C):
    from mixdq_amd.nn.glue import HipAttnProcessor, swap_glue_modules, unswap_glue_modules
    from mixdq_amd.unet import Attention
    torch.manual_seed(0)
    a = Attention(640, 2048, 64).half().to(DEV)              # cross-attention (77 keys: the short-key kernel)
    a_self = Attention(640, None, 64).half().to(DEV)         # self-attention (the pipelined kernel)
    x = t(dd.normal_f16(81, (2, 1024, 640), 1.0))
    ctx = t(dd.normal_f16(82, (2, 77, 2048), 1.0))
    with torch.no_grad():
        stock_self, stock_cross = a_self(x), a(x, ctx)
        assert swap_glue_modules(nn.ModuleList([a, a_self]))["attention"] == 2 and isinstance(a, Attention)
        got_self, got_cross = a_self(x), a(x, ctx)
        q, k, v = a.to_q(x), a.to_k(ctx), a.to_v(ctx)
        want = a.to_out[0](C.attention_f16(q, k, v, a.heads))
    assert torch.equal(got_cross.view(torch.int16), want.view(torch.int16))
    for got, stock in ((got_self, stock_self), (got_cross, stock_cross)):      # the tolerance of the kernel's own tests
        assert ((got.float() - stock.float()).abs() <= 4e-3 + 8e-3 * stock.float().abs()).all()

    class FakeDiffusersAttention(nn.Module):
        """The attributes of diffusers.models.attention_processor.Attention a processor touches."""

        def __init__(self, src):
            super().__init__()
            self.to_q, self.to_k, self.to_v, self.to_out = src.to_q, src.to_k, src.to_v, src.to_out
            self.heads, self.processor = src.heads, None
            self.group_norm = self.spatial_norm = self.norm_q = self.norm_k = None
            self.norm_cross, self.residual_connection, self.rescale_output_factor = False, False, 1.0

        def set_processor(self, p):
            self.processor = p

        def forward(self, hidden_states, encoder_hidden_states=None, **kw):
            return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)

    unswap_glue_modules(nn.ModuleList([a]))
    fake = FakeDiffusersAttention(a).to(DEV)
    stock_proc = lambda attn, hs, encoder_hidden_states=None, attention_mask=None, temb=None: "stock"   # noqa: E731
    fake.set_processor(stock_proc)
    assert swap_glue_modules(nn.ModuleList([fake]))["attention"] == 1 and isinstance(fake.processor, HipAttnProcessor)
    with torch.no_grad():
        assert torch.equal(fake(x, ctx).view(torch.int16), want.view(torch.int16))
        assert fake(x, ctx, attention_mask=torch.zeros(1, device=DEV)) == "stock"      # masks: the replaced processor
    unswap_glue_modules(nn.ModuleList([fake]))
    assert fake.processor is stock_proc


@pytest.mark.parametrize("heads64", [False, True], ids=["heads16", "heads64"])
def test_swapped_unet_is_the_chain_of_its_kernels(C, heads64):
    """The tiny UNet, Linear / Conv2d swapped (the reference's surface) and then swap_glue=True: same names and
    state_dict keys, output == the de-fused reference of the fused graph (every fused launch replaced by the chain of
    this repo's FP16-output kernels + the layer's own quantize launch: the same arithmetic at every rounding point
    except the residual adds, which stay torch half adds on both sides), hipGraph replay == eager, and the swap
    undone gives the stock drop-in network's bits back."""
    import bench
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.nn.glue import unswap_glue_modules
    from mixdq_amd.quantize_sdxl import example_inputs, hip_graph_opt, quantize_unet
    from mixdq_amd.unet import build_unet, defused, quantizable_layers
    # heads64: 64-wide heads and 256 / 64 tokens per level -- the attention modules then run this repo's kernels and
    # the launch forms of _attention_hand_off (one q | k | v GEMM, to_q + cross-attention + quantize in one launch);
    # heads16: PyTorch's SDPA stays (mixdq_attention_f16 is head_dim 64 only), everything else is swapped
    cfg = dict(bench.TINY_CFG, block_out_channels=(64, 128, 256), head_dim=64) if heads64 else bench.TINY_CFG
    unet = build_unet(DEV, cfg=cfg)
    inputs = example_inputs(2, 32 if heads64 else 16, DEV, seed=7)
    ckpt = calibrate(unet, [inputs])
    bos_dict = precompute_bos(unet, inputs["encoder_hidden_states"])
    names = list(quantizable_layers(unet))
    with torch.no_grad():
        stock_fp16 = unet(**inputs)[0].float()
    quantize_unet(unet, bench.Cfg({n: 8 for n in names}, {n: 8 for n in names if n not in ("conv_in", "conv_out")}),
                  ckpt, bos=True, bos_dict=bos_dict)
    keys = list(unet.state_dict())
    with torch.no_grad():
        dropin = unet(**inputs)[0].clone()
    from mixdq_amd.nn.glue import swap_glue_modules
    n = swap_glue_modules(unet, operands=False)        # every layer still runs its own quantize launch
    assert n["groupnorm"] == 46 and n["silu_folded"] == 35 and n["layernorm"] > 0 and n["geglu"] > 0 and n["attention"] > 0
    assert n["operand_links"] == 0 and n["attention_handoff"] == 0
    with torch.no_grad():
        no_handoff = unet(**inputs)[0].clone()
        k_no = bench.count_kernels(lambda: unet(**inputs), torch.device(DEV))
    n = swap_glue_modules(unet)                        # ... and with the producers' INT8 operands handed on
    assert n["operand_links"] > 0 and n["attention_handoff"] > 0 and n["groupnorm"] == 0
    assert list(unet.state_dict()) == keys
    with torch.no_grad():
        glue = unet(**inputs)[0].clone()
        k_yes = bench.count_kernels(lambda: unet(**inputs), torch.device(DEV))
    assert torch.equal(glue.view(torch.int16), no_handoff.view(torch.int16))     # the same bits, fewer launches:
    n_tb = sum(1 for m in unet.modules() if type(m).__name__ == "BasicTransformerBlock")
    # per block: the quantize launches of q | k | v, to_q, the GEGLU projection, net.2, two to_out.0 and one of the two
    # of attn2.to_k / to_v (equal quantizers: the context is quantized once); per ResNet block those of conv1 / conv2
    # (heads64) ... and two launches of the self-attention's three projections, one of to_q / cross-attention
    assert k_no - k_yes >= (12 if heads64 else 6) * n_tb, (k_no, k_yes)
    if heads64:
        from mixdq_amd.nn.glue import _HipAttend
        assert all("_qkv" in m.__dict__ for m in unet.modules()
                   if isinstance(m, _HipAttend) and m.to_k.in_features == m.to_q.in_features)
    with torch.no_grad():
        unet.set_fused(True)
        with defused():
            ref = unet(**inputs)[0].clone()
        unet.set_fused(False)
    assert torch.equal(glue.view(torch.int16), ref.view(torch.int16)), \
        f"{(glue.view(torch.int16) != ref.view(torch.int16)).sum().item()} of {glue.numel()} differ"
    # within quantization noise of the stock drop-in network (FP16 norms differ in the last bit; INT8 steps flip)
    noise = (dropin.float() - stock_fp16).abs().mean()
    assert (glue.float() - dropin.float()).abs().mean() <= 1.5 * noise
    hip_graph_opt(unet)
    with torch.no_grad():
        for _ in range(2):
            assert torch.equal(unet(**inputs)[0].view(torch.int16), glue.view(torch.int16))
    unet.forward = unet.forward.__wrapped__
    unswap_glue_modules(unet)
    with torch.no_grad():
        assert torch.equal(unet(**inputs)[0].view(torch.int16), dropin.view(torch.int16))



class _Lin(nn.Module):
    """What an operand hand-off looks at in a consumer: its two quantizer buffers, its width, that it is accelerated."""
    valid_for_acceleration = True

    def __init__(self, k, scale, zp):
        super().__init__()
        self.in_features = k
        self.register_buffer("act_scales_inv", torch.tensor(1.0 / scale))
        self.register_buffer("act_zero_points", torch.tensor(float(zp)))


class BasicTransformerBlock(nn.Module):          # (the class NAME is what OPERAND_PAIRS keys on)
    def __init__(self, c):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(c), nn.LayerNorm(c), nn.LayerNorm(c)
        self.attn1 = nn.Module()
        self.attn1.to_q, self.attn1.to_k, self.attn1.to_v = _Lin(c, 0.05, 3), _Lin(c, 0.05, 3), _Lin(c, 0.031, -7)
        self.attn2 = nn.Module()
        self.attn2.to_q = _Lin(c, 0.04, 0)


def test_operand_hand_off_is_quantize_of_the_fp16_output_and_only_for_that_tensor(C):
    """A swapped LayerNorm with known consumers: one INT8 tensor per distinct quantizer, each == the quantize launch
    applied to the FP16 tensor it returns; found only on THAT tensor object, unmodified, for a linked quantizer."""
    from mixdq_amd.nn.glue import swap_glue_modules, tagged_operand, unswap_glue_modules
    c = 640
    blk = BasicTransformerBlock(c).half().to(DEV)
    for m in blk.modules():
        if isinstance(m, _Lin):
            m.float()
    with torch.no_grad():
        blk.norm1.weight.copy_(t((dd.normal_f16(71, (c,), 0.3).astype(np.float32) + 1).astype(np.float16)))
        blk.norm1.bias.copy_(t(dd.normal_f16(72, (c,), 0.2)))
    x = t(dd.normal_f16(73, (2, 77, c), 1.5))
    n = swap_glue_modules(blk)
    assert n["layernorm"] == 3 and n["operand_links"] == 4           # (norm3 -> ff.net.0.proj: no such module here)
    y = blk.norm1(x)
    a = blk.attn1
    qq, qk, qv = (tagged_operand(y, m) for m in (a.to_q, a.to_k, a.to_v))
    assert qq is not None and qq is qk and qv is not None and qv is not qq       # equal quantizers share one tensor
    for m, q in ((a.to_q, qq), (a.to_v, qv)):
        want = C.quantize_per_tensor_to_int8(y, m.act_scales_inv, m.act_zero_points)
        assert q.dtype == torch.int8 and torch.equal(q, want)
    assert torch.equal(y, C.layernorm_quantize(x, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, [], want_f16=True)[1])
    assert tagged_operand(y, blk.attn2.to_q) is None                  # another producer's consumer
    assert tagged_operand(y.clone(), a.to_q) is None and tagged_operand(y[:1], a.to_q) is None
    assert tagged_operand(y.view(-1, c), a.to_q) is None              # a view is another tensor object
    other = _Lin(c, 0.05, 3).to(DEV)
    assert tagged_operand(y, other) is None                           # equal values, another layer's buffers
    y.add_(0)                                                         # modified in place since: stale
    assert tagged_operand(y, a.to_q) is None
    with torch.no_grad():
        a.to_k.act_scales_inv.fill_(1 / 0.02)                         # a re-calibrated consumer: regrouped
    y = blk.norm1(x)
    qq, qk = tagged_operand(y, a.to_q), tagged_operand(y, a.to_k)
    assert qk is not qq and torch.equal(qk, C.quantize_per_tensor_to_int8(y, a.to_k.act_scales_inv, a.to_k.act_zero_points))
    a.to_v.valid_for_acceleration = False                             # an FP16 fallback layer: takes the FP16 tensor
    assert tagged_operand(blk.norm1(x), a.to_v) is None
    unswap_glue_modules(blk)
    assert tagged_operand(blk.norm1(x), a.to_q) is None
