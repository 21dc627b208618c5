"""GPU tests of the FP16 layer kernels (mixdq_linear_f16 / mixdq_conv2d_f16): the reference's FP
fallback for layers without an activation quantizer (nn/Linear.py:155-156, nn/Conv2d.py:306-309 --
F.linear / F.conv2d there) on this repo's MFMA kernel family.  Floating point, so the oracle is a
plain PyTorch reference of the same op evaluated in FP64 on the CPU (MIOpen's FP32 convolutions
may themselves be Winograd-transformed) and the bar a tolerance, stated here:

    |out - ref| <= 2^-10 * |ref| + 2^-10 * rms(ref)

i.e. one fp16 rounding of the exact result (relative half-ulp = 2^-11) plus the FP32 accumulation
error, with an absolute floor for outputs that cancel to ~0.  PyTorch's own FP16 op (hipBLASLt /
MIOpen) is held to the same bound as a sanity check of the bound itself."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(out, ref):
    ref = ref.to(out.device).float()
    tol = 2.0 ** -10 * ref.abs() + 2.0 ** -10 * ref.pow(2).mean().sqrt()
    err = (out.float() - ref).abs()
    bad = err > tol
    assert not bad.any(), f"{int(bad.sum())} of {bad.numel()} outside tolerance; max err " \
                          f"{err.max().item():.3e} (tol there {tol.flatten()[err.argmax()].item():.3e})"


def rnd(shape, seed, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).half().to(DEV)


LIN = [  # M, K, N, bias
    (1024, 5120, 1280, True),      # the act-protected ff.net.2 layers of act_8.00.yaml
    (203, 1232, 136, True),        # ragged M / N, K % 64 != 0 (general staging path)
    (331, 1280, 424, False),       # fast staging path with M / N tails
    (77, 2048, 640, False), (1, 1280, 1280, True), (4096, 640, 640, True), (5, 8, 4, True),
]


@pytest.mark.parametrize("M,K,N,bias", LIN)
def test_linear_f16_vs_fp32_reference(C, M, K, N, bias):
    x, w = rnd((M, K), 1), rnd((N, K), 2, 0.05)
    b = rnd((N,), 3) if bias else None
    ref = F.linear(x.cpu().double(), w.cpu().double(), None if b is None else b.cpu().double())
    out = C.linear_f16(x, w, b)
    assert out.dtype == torch.float16 and tuple(out.shape) == (M, N)
    close(out, ref)
    close(F.linear(x, w, b), ref)                    # the bound holds for PyTorch's FP16 op too
    assert torch.equal(out, C.linear_f16(x, w, b))   # deterministic


@pytest.mark.parametrize("cfg", __import__("mixdq_amd._C", fromlist=["x"]).F16_CONFIGS
                         if torch.cuda.is_available() else [])
def test_linear_and_conv_f16_every_configuration(C, cfg):
    for (M, K, N) in ((203, 1232, 136), (331, 1280, 424)):
        x, w, b = rnd((M, K), 4), rnd((N, K), 5, 0.05), rnd((N,), 6)
        close(C.linear_f16(x, w, b, _cfg=cfg),
              F.linear(x.cpu().double(), w.cpu().double(), b.cpu().double()))
    x = rnd((2, 64, 13, 11), 7).contiguous(memory_format=torch.channels_last)
    w, b = rnd((72, 64, 3, 3), 8, 0.05), rnd((72,), 9)
    close(C.conv2d_f16(x, w, b, 2, 1, _cfg=cfg),
          F.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), 2, 1))


def test_every_f16_configuration_accumulates_in_the_same_order(C):
    """One MFMA shape, no k-split, k ascending in every FP16 tile: the results agree bit for bit, so
    the tile rule (which looks at M = batch x rows) cannot change an image with the batch size."""
    x, w, b = rnd((331, 2560), 14), rnd((424, 2560), 15, 0.05), rnd((424,), 16)
    outs = [C.linear_f16(x, w, b, _cfg=cfg) for cfg in C.F16_CONFIGS]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert torch.equal(C.linear_f16(x[:7], w, b), outs[0][:7])          # ... nor with M
    xc = rnd((2, 64, 13, 11), 17).contiguous(memory_format=torch.channels_last)
    wc, bc = rnd((72, 64, 3, 3), 18, 0.05), rnd((72,), 19)
    outs = [C.conv2d_f16(xc, wc, bc, 1, 1, _cfg=cfg) for cfg in C.F16_CONFIGS]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert torch.equal(C.conv2d_f16(xc[:1], wc, bc, 1, 1), outs[0][:1])


def test_linear_f16_leading_dims_residual_and_errors(C):
    x, w, b = rnd((2, 77, 320), 10), rnd((640, 320), 11, 0.05), rnd((640,), 12)
    res = rnd((2, 77, 640), 13)
    out = C.linear_f16(x, w, b, _residual=res)
    assert tuple(out.shape) == (2, 77, 640)
    # the residual is added AFTER the fp16 rounding, exactly as a following torch half add
    assert torch.equal(out, C.linear_f16(x, w, b) + res)
    assert tuple(C.linear_f16(x[:0], w, b).shape) == (0, 77, 640)
    with pytest.raises(RuntimeError, match="fp16 GPU tensor"):
        C.linear_f16(x.float(), w, b)
    with pytest.raises(RuntimeError, match="last dimension"):
        C.linear_f16(x[..., :8], w, b)


CONV = [  # N, C, H, W, K, ksize, stride, pad, bias
    (1, 4, 32, 32, 320, 3, 1, 1, True),        # conv_in: 4 input channels -> the generic kernel
    (1, 320, 32, 32, 4, 3, 1, 1, True),        # conv_out: 4 output channels
    (2, 320, 24, 24, 320, 3, 1, 1, True),      # down_blocks.0.resnets.0.conv2 (act-protected)
    (1, 640, 16, 16, 320, 1, 1, 0, True),      # a 1x1 shortcut
    (2, 64, 13, 11, 72, 3, 2, 1, False),       # stride 2, odd sizes
    (1, 128, 5, 5, 64, 3, 1, 1, True), (3, 8, 7, 9, 12, 3, 1, 0, True),
]


@pytest.mark.parametrize("case", CONV, ids=[f"c{c[1]}_k{c[4]}_{c[5]}x{c[5]}_s{c[6]}p{c[7]}" for c in CONV])
def test_conv2d_f16_vs_fp32_reference(C, case):
    N, Cin, H, W, K, ks, stride, pad, bias = case
    x = rnd((N, Cin, H, W), 20).contiguous(memory_format=torch.channels_last)
    w = rnd((K, Cin, ks, ks), 21, 0.05)
    b = rnd((K,), 22) if bias else None
    ref = F.conv2d(x.cpu().double(), w.cpu().double(), None if b is None else b.cpu().double(),
                   stride, pad)
    out = C.conv2d_f16(x, w, b, stride, pad)
    assert out.shape == ref.shape and out.dtype == torch.float16
    assert out.is_contiguous(memory_format=torch.channels_last) or out.shape[1] == 1 or \
        out.shape[2] * out.shape[3] == 1
    close(out, ref)
    assert torch.equal(out, C.conv2d_f16(x.contiguous(), w.contiguous(), b, stride, pad))  # NCHW in


def test_conv2d_f16_residual_forms(C):
    x = rnd((2, 64, 12, 12), 30).contiguous(memory_format=torch.channels_last)
    w, b = rnd((96, 64, 3, 3), 31, 0.05), rnd((96,), 32)
    base = C.conv2d_f16(x, w, b, 1, 1)
    full = rnd((2, 96, 12, 12), 33).contiguous(memory_format=torch.channels_last)
    assert torch.equal(C.conv2d_f16(x, w, b, 1, 1, _residual=full), base + full)
    per_img = rnd((2, 96), 34)
    assert torch.equal(C.conv2d_f16(x, w, b, 1, 1, _residual=per_img, _residual_per_image=True),
                       base + per_img[:, :, None, None])


def test_fp_fallback_modules_run_on_the_fp16_kernels(C, modules_golden):
    """QuantizedLinear / QuantizedConv2d without an activation quantizer (the reference's FP
    fallback) give F.linear / F.conv2d within the FP16 bound -- on this repo's kernels."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from tests.cases import MODULE_CASES, module_ckpt, module_input
    from tests.test_host import prepared
    for c in MODULE_CASES:
        if c.get("bos") or c.get("split"):
            continue
        cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
        fm = prepared(c, modules_golden, with_act=False)
        qm = cls.from_float(fm, ckpt=module_ckpt(c, modules_golden)).to(DEV)
        assert not qm.valid_for_acceleration and qm.fp16_kernel
        x = module_input(c).to(DEV)
        with torch.no_grad():
            y = qm(x)
            ref = fm.double()(x.cpu().double())
        close(y, ref)
