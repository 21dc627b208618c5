"""GPU parity tests of the five operator entry points: HIP path (through the C-ABI) vs the
oracle and vs the committed golden hashes, bit-exact.  Mirrors the reference's self-tests
op/quant.py:7-30, op/qlinear.py:28-108, op/qconv2d.py:25-119 (same shapes, seeded)."""
import hashlib

import numpy as np
import pytest
import torch

from tests import detdata as dd
from tests.cases import LINEAR_CASES, CONV_CASES, linear_inputs, conv_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["A", "B"], ids=["fma", "mul_add"])
def epilogue_variant(request, monkeypatch):
    """Every test of this file runs under both roundings of the multiply-adds (SURVEY.md Appendix
    B): A = one FMA (what nvcc -fmad=true makes of the reference's mul + add; the default), B =
    multiply, round, add (MIXDQ_FLAG_UNFUSED; `MIXDQ_EPILOGUE_VARIANT=B`).  Which of the two the
    reference's CUDA binary used cannot be observed here, so both stay pinned to the oracle."""
    import mixdq_amd._C as C_
    monkeypatch.setattr(C_, "FLAGS", 1 if request.param == "B" else 0)
    return request.param


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x if dtype is None else x.to(dtype)


def scal(v):
    return torch.tensor(float(v), dtype=torch.float32, device=DEV)


def assert_bits_equal(got: np.ndarray, want: np.ndarray, what: str):
    g = got.view(np.uint16) if got.dtype == np.float16 else got
    w = want.view(np.uint16) if want.dtype == np.float16 else want
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    bad = np.nonzero(g.reshape(-1) != w.reshape(-1))[0]
    if bad.size:
        i = bad[0]
        raise AssertionError(f"{what}: {bad.size}/{g.size} elements differ; first at flat index "
                             f"{i}: got {got.reshape(-1)[i]!r} want {want.reshape(-1)[i]!r}")


# ------------------------------------------------------------------------------ quantize (a1)
def test_quantize_reference_case(C, oracle, ops_golden, ops_small):
    """op/quant.py:24-27: kernel == torch.quantize_per_tensor == vectorized kernel."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_ref")
    x = dd.f16(case["seed"], tuple(case["shape"]))
    xd = t(x)
    q1 = C.quantize_per_tensor_to_int8(xd, scal(case["scale_inv"]), scal(case["zp"]))
    q2 = C.quantize_per_tensor_to_int8_vectorized(xd, scal(case["scale_inv"]), scal(case["zp"]))
    assert q1.dtype == torch.int8 and q1.shape == xd.shape
    assert_bits_equal(q1.cpu().numpy(), ops_small[case["expect"]], "q_ref vs torch golden")
    assert torch.equal(q1, q2)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_quantize_activation_ranges(C, oracle, ops_golden, i):
    case = next(c for c in ops_golden["quantize"] if c["name"] == f"q_act{i}")
    x = dd.normal_f16(case["seed"], tuple(case["shape"]), std=case["std"])
    q = C.quantize_per_tensor_to_int8(t(x), scal(case["scale_inv"]), scal(case["zp"]))
    want = oracle.quantize(x, case["scale_inv"], case["zp"], C.FLAGS & 1)
    assert_bits_equal(q.cpu().numpy(), want, case["name"])
    assert sha(q.cpu().numpy()) == case["sha_B" if C.FLAGS & 1 and "sha_B" in case else "sha_A"]


@pytest.mark.parametrize("name", ["q_edge_pow2", "q_edge_pow2_zp", "q_edge_half"])
def test_quantize_edges(C, ops_golden, ops_small, name):
    """ties-to-even, saturation, signed zero, subnormals, +-inf."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == name)
    x = ops_small[case["x"]]
    q = C.quantize_per_tensor_to_int8(t(x), scal(case["scale_inv"]), scal(case["zp"]))
    assert_bits_equal(q.cpu().numpy(), ops_small[case["expect"]], name)


def test_quantize_fma_separator(C, oracle, ops_golden, ops_small):
    """(x, s_inv, zp) triples where fma(x, s, zp) and (x*s)+zp round differently: the HIP build
    follows the selected variant (default A = fused)."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_sep")
    x, si, zp = ops_small[case["x"]], ops_small[case["scale_inv"]], ops_small[case["zp"]]
    want = ops_small[case["expect_B" if C.FLAGS & 1 else "expect_A"]]
    for i in range(x.size):
        q = C.quantize_per_tensor_to_int8(t(x[i:i + 1]), scal(si[i]), scal(zp[i]))
        assert int(q.item()) == int(want[i]), (i, x[i], si[i], zp[i])
    # every finite half value, both signs, one realistic scale
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_allhalf")
    allh = np.arange(1, 0x7c00, dtype=np.uint16).view(np.float16)
    allh = np.concatenate([allh, -allh])
    q = C.quantize_per_tensor_to_int8(t(allh), scal(case["scale_inv"]), scal(case["zp"]))
    assert sha(q.cpu().numpy()) == case["sha_B" if C.FLAGS & 1 else "sha_A"]


def test_quantize_strided_bos_slice(C, ops_golden, ops_small):
    """x[:, 1:, :] (nn/Linear.py:180): intended strided semantics, any batch size."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_bos_slice")
    x = t(dd.normal_f16(case["seed"], tuple(case["shape"])))
    q = C.quantize_per_tensor_to_int8(x[:, 1:, :], scal(case["scale_inv"]), scal(case["zp"]))
    assert q.shape == (2, 76, 64)
    assert_bits_equal(q.cpu().numpy(), ops_small[case["expect"]], "bos slice")


@pytest.mark.parametrize("channels_last", [False, True])
def test_quantize_strided_channel_split(C, ops_golden, ops_small, channels_last):
    """x[:, :split], x[:, split:] (nn/Conv2d.py:313-316), NCHW and channels-last, batch 2."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_chan_split")
    x = t(dd.normal_f16(case["seed"], tuple(case["shape"])))
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    s = case["split"]
    lo = C.quantize_per_tensor_to_int8(x[:, :s], scal(case["scale_inv"][0]), scal(case["zp"][0]))
    hi = C.quantize_per_tensor_to_int8(x[:, s:], scal(case["scale_inv"][1]), scal(case["zp"][1]))
    assert_bits_equal(lo.cpu().numpy(), ops_small[case["expect"][0]], "split lo")
    assert_bits_equal(hi.cpu().numpy(), ops_small[case["expect"][1]], "split hi")


@pytest.mark.parametrize("shape", [(0,), (1,), (7,), (8,), (1027,), (3, 5, 7), (2, 0, 4)])
def test_quantize_ragged_and_empty(C, oracle, shape):
    x = dd.normal_f16(77, shape, std=2.0) if int(np.prod(shape)) else np.zeros(shape, np.float16)
    q = C.quantize_per_tensor_to_int8(t(x), scal(10.5), scal(-3.0))
    assert q.shape == tuple(shape)
    if x.size:
        assert_bits_equal(q.cpu().numpy(), oracle.quantize(x, 10.5, -3.0, C.FLAGS & 1), str(shape))


def test_quantize_large_roundtrip_property(C):
    """Full-size property (1024 px level-0 activation, 16384 x 320): dequantising recovers x to
    within half a step wherever the value was not clipped."""
    torch.manual_seed(0)
    x = (torch.randn(16384, 320, device=DEV) * 1.5).half()
    s, zp = 0.0312, -11.0
    q = C.quantize_per_tensor_to_int8(x, scal(1.0 / np.float32(s)), scal(zp))
    deq = (q.float() - zp) * s
    inside = (q > -128) & (q < 127)
    err = (deq - x.float()).abs()[inside].max().item()
    assert err <= 0.5 * s * 1.001


def test_quantize_errors(C):
    x = torch.zeros(8, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="input should be on CUDA"):
        C.quantize_per_tensor_to_int8(x.cpu(), scal(1), scal(0))
    with pytest.raises(RuntimeError, match="input should be fp16"):
        C.quantize_per_tensor_to_int8(x.float(), scal(1), scal(0))
    with pytest.raises(RuntimeError, match="scale_inv should be fp32"):
        C.quantize_per_tensor_to_int8(x, scal(1).half(), scal(0))
    with pytest.raises(RuntimeError, match="zero_point should be fp32"):
        C.quantize_per_tensor_to_int8(x, scal(1), scal(0).half())


# ------------------------------------------------------------------------------- qlinear (a2)
def run_qlinear(C, case):
    a, w, wscale, in_scale, in_zp, bias, scale, bias0 = linear_inputs(case)
    wsum = w.astype(np.float32).sum(axis=1, dtype=np.float32)
    out = C.qlinear_w8_a8_ohalf(t(a), t(w), t(wscale), scal(in_scale), scal(in_zp), t(wsum),
                                t(scale), t(bias0), None if bias is None else t(bias))
    return out, (a, w, bias0, scale, bias)


@pytest.mark.parametrize("case", LINEAR_CASES, ids=[c[0] for c in LINEAR_CASES])
def test_qlinear_bit_exact(C, oracle, ops_golden, case):
    out, (a, w, bias0, scale, bias) = run_qlinear(C, case)
    g = next(c for c in ops_golden["qlinear"] if c["name"] == case[0])
    assert out.dtype == torch.float16 and out.is_contiguous()
    assert tuple(out.shape) == (g["M"], g["N"])
    variant = C.FLAGS & 1
    want = oracle.qlinear(a, w, bias0, scale, bias, variant)
    assert_bits_equal(out.cpu().numpy(), want, case[0])
    assert sha(out.cpu().numpy()) == g["sha_B" if variant else "sha_A"], "golden hash drift"


def test_qlinear_reference_tolerances(C):
    """op/qlinear.py:66-101 restated: kernel vs integer reference (atol 1e-4, rtol 1e-2) and vs
    FP reference (rtol 1e-2, atol 1e-2), the reference's run_test(64, 8, 16)."""
    case = LINEAR_CASES[0]
    out, (a, w, bias0, scale, bias) = run_qlinear(C, case)
    _, _, wscale, in_scale, in_zp, *_ = linear_inputs(case)
    ai, wi = t(a), t(w)
    ws, bs = t(wscale), t(bias)
    infused = ws * float(in_scale)
    offset = ws * wi.to(torch.int32).sum(dim=1) * (float(in_zp) * float(in_scale))
    gemm = torch.matmul(ai.float(), wi.float().t())
    ref_int = (gemm * infused - offset + bs.float()).half()
    ref_fp = (torch.matmul((ai.float() - float(in_zp)) * float(in_scale),
                           (wi.float() * ws[:, None]).t()) + bs.float()).half()
    torch.testing.assert_close(out, ref_int, atol=1e-4, rtol=1e-2)
    torch.testing.assert_close(out, ref_fp, atol=1e-2, rtol=1e-2)


def test_qlinear_leading_dims_and_noncontiguous(C, oracle):
    a = dd.int8(11, (2, 5, 64))
    w = dd.int8(12, (3, 48, 64))[1]          # a view; made contiguous inside (qlinear.cc:75-77)
    b0, sc = dd.f32(13, (48,), -50, 50), dd.f32(14, (48,), 0.001, 0.01)
    at = t(a).transpose(0, 1)                # non-contiguous input
    out = C.qlinear_w8_a8_ohalf(at, t(dd.int8(12, (3, 48, 64)))[1], t(sc), scal(1), scal(0),
                                t(b0), t(sc), t(b0), None)
    assert tuple(out.shape) == (5, 2, 48)
    want = oracle.qlinear(np.ascontiguousarray(a.transpose(1, 0, 2)), w, b0, sc, None, C.FLAGS & 1)
    assert_bits_equal(out.cpu().numpy(), want, "leading dims")


def test_qlinear_row_map(C, oracle):
    """BOS row map (include/mixdq_hip.h mixdq_qlinear_w8a8_rows): rows 1..76 of a [B,77,N]."""
    B, T, K, N = 2, 77, 64, 40
    a = dd.int8(21, (B, T - 1, K))
    w = dd.int8(22, (N, K))
    b0, sc = dd.f32(23, (N,), -50, 50), dd.f32(24, (N,), 0.001, 0.01)
    out = torch.full((B, T, N), 7.0, dtype=torch.float16, device=DEV)
    C.qlinear_w8_a8_ohalf(t(a), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), None,
                          _out=out, _row_map=(T - 1, T, 1))
    want = oracle.qlinear(a, w, b0, sc, None, C.FLAGS & 1)
    assert_bits_equal(out[:, 1:].cpu().numpy(), want, "row map body")
    assert (out[:, 0] == 7.0).all()


def test_qlinear_full_size_properties(C):
    """BASELINE size (1024 px, M = 4096 x N = 5120 x K = 640): linearity in the accumulator --
    out(a1) + out(a2) - out(0) == out(a1 + a2) exactly in the integer domain (scale 1, bias0 0,
    values small enough that f16 is exact) -- and a row-permutation equivariance."""
    M, K, N = 4096, 640, 5120
    g = torch.Generator(device="cpu").manual_seed(5)
    a1 = torch.randint(-2, 3, (M, K), generator=g, dtype=torch.int8).to(DEV)
    a2 = torch.randint(-2, 3, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-1, 2, (N, K), generator=g, dtype=torch.int8).to(DEV)
    one = torch.ones(N, device=DEV)
    zero = torch.zeros(N, device=DEV)

    def f(a):
        return C.qlinear_w8_a8_ohalf(a, w, one, scal(1), scal(0), zero, one, zero, None)

    o1, o2, o12 = f(a1), f(a2), f(a1 + a2)
    assert torch.equal(o1.float() + o2.float(), o12.float())     # |acc| <= 2560 < 2048*2: exact
    ref = (a1.float() @ w.float().t())
    assert torch.equal(o1.float(), ref)
    perm = torch.randperm(M, generator=g).to(DEV)
    assert torch.equal(f(a1[perm]), o1[perm])


def test_qlinear_errors(C):
    a = torch.zeros(4, 16, dtype=torch.int8, device=DEV)
    w = torch.zeros(8, 16, dtype=torch.int8, device=DEV)
    v = torch.zeros(8, device=DEV)
    s0 = scal(0)
    with pytest.raises(RuntimeError, match="Input should be on GPU"):
        C.qlinear_w8_a8_ohalf(a.cpu(), w, v, s0, s0, v, v, v)
    with pytest.raises(RuntimeError, match="input_int8 should be int8 type"):
        C.qlinear_w8_a8_ohalf(a.float(), w, v, s0, s0, v, v, v)
    with pytest.raises(RuntimeError, match="last dimension of input and weight should match"):
        C.qlinear_w8_a8_ohalf(a[:, :8], w, v, s0, s0, v, v, v)
    with pytest.raises(RuntimeError, match="weight_scale vector should be equal"):
        C.qlinear_w8_a8_ohalf(a, w, v[:4], s0, s0, v, v, v)
    with pytest.raises(RuntimeError, match="bias with float16"):
        C.qlinear_w8_a8_ohalf(a, w, v, s0, s0, v, v, v, v)
    a6 = torch.zeros(4, 6, dtype=torch.int8, device=DEV)
    w6 = torch.zeros(8, 6, dtype=torch.int8, device=DEV)
    with pytest.raises(RuntimeError, match="alignment not to 4 is not supported"):
        C.qlinear_w8_a8_ohalf(a6, w6, v, s0, s0, v, v, v)


def test_qlinear_empty(C):
    a = torch.zeros(0, 16, dtype=torch.int8, device=DEV)
    w = torch.zeros(8, 16, dtype=torch.int8, device=DEV)
    v = torch.zeros(8, device=DEV)
    out = C.qlinear_w8_a8_ohalf(a, w, v, scal(0), scal(0), v, v, v)
    assert tuple(out.shape) == (0, 8)


# ------------------------------------------------------------------------------- qconv2d (a3/a4)
def run_qconv(C, case):
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    xin = t(x).permute(0, 3, 1, 2)            # NCHW-shaped view of NHWC memory (= channels_last)
    win = t(wt).permute(0, 3, 1, 2)
    out = C.qconv2d_w8_a8_ohalf(
        xin, win, t(wscale), scal(in_scale), scal(in_zp), t(scale),
        t(wsum.reshape(k, 1, r, s)) if pad > 0 else None,
        t(bias0) if pad == 0 else None,
        None if bias is None else t(bias), stride, pad)
    return out, (x, wt, scale, wsum, in_zp, bias0, bias)


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_qconv2d_bit_exact(C, oracle, ops_golden, case):
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    out, (x, wt, scale, wsum, in_zp, bias0, bias) = run_qconv(C, case)
    g = next(cc for cc in ops_golden["qconv2d"] if cc["name"] == name)
    assert out.dtype == torch.float16
    assert out.is_contiguous(memory_format=torch.channels_last) or out.shape[1] == 1 or \
        out.shape[2] * out.shape[3] == 1
    variant = C.FLAGS & 1
    want = oracle.qconv2d(x, wt, scale, wsum if pad > 0 else None, in_zp,
                          bias0 if pad == 0 else None, bias, stride, pad, variant)
    got = out.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    assert_bits_equal(got, want, name)
    assert sha(got) == g["sha_B" if variant else "sha_A"], "golden hash drift"


def test_qconv2d_nchw_input_is_converted(C, oracle):
    """qconv2d.cc:91-95: a contiguous-NCHW int8 input is converted to channels-last internally."""
    case = next(c for c in CONV_CASES if c[0] == "conv_odd_hw")
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    xin = t(x).permute(0, 3, 1, 2).contiguous()          # true NCHW memory
    win = t(wt).permute(0, 3, 1, 2).contiguous()
    out = C.qconv2d_w8_a8_ohalf(xin, win, t(wscale), scal(in_scale), scal(in_zp), t(scale),
                                t(wsum.reshape(k, 1, r, s)), None, None, stride, pad)
    want = oracle.qconv2d(x, wt, scale, wsum, in_zp, None, None, stride, pad, C.FLAGS & 1)
    assert_bits_equal(out.permute(0, 2, 3, 1).contiguous().cpu().numpy(), want, "nchw input")


def test_zero_point_propagate_matches_reference_formula(C, oracle):
    """a4: materialised bias0 == oracle restatement of conv_act_zero_point_propagate.cu, and the
    cached border table reproduces it."""
    k, r, s, n, h, w_, stride, pad = 24, 3, 3, 2, 5, 7, 2, 1
    wsum = dd.f32(31, (k, 1, r, s), -2000, 2000).round()
    zp = scal(-13.0)
    got = C.conv_zero_point_propagate(t(wsum), zp, n, h, w_, stride, pad)
    want = oracle.zp_propagate(wsum, -13.0, n, h, w_, stride, pad)
    assert_bits_equal(got.permute(0, 2, 3, 1).contiguous().cpu().numpy(), want, "zp propagate")
    table = C.conv_border_table(t(wsum)).cpu().numpy()
    # class ((rlo*R+rhi)*S+slo)*S+shi: full window == sum of all taps
    full = ((0 * r + (r - 1)) * s + 0) * s + (s - 1)
    assert np.array_equal(table[full], wsum.reshape(k, -1).sum(axis=1, dtype=np.float32))


def test_qconv2d_full_size_properties(C):
    """BASELINE size (1024 px level 0: 128 x 128 x 320 -> 320, 3x3 p1): an all-ones kernel on an
    all-ones image counts the in-bounds taps of every pixel (9 / 6 / 4), and translation
    equivariance holds away from the border."""
    H = W = 128
    Cin = Cout = 320
    x = torch.ones(1, Cin, H, W, dtype=torch.int8, device=DEV).contiguous(
        memory_format=torch.channels_last)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.int8, device=DEV).contiguous(
        memory_format=torch.channels_last)
    w[:, 0] = 1                                         # only input channel 0 contributes
    one = torch.ones(Cout, device=DEV)
    wsum = w.float().sum(dim=1, keepdim=True)
    out = C.qconv2d_w8_a8_ohalf(x, w, one, scal(1), scal(0), one, wsum, None, None, 1, 1)
    cnt = torch.nn.functional.conv2d(torch.ones(1, 1, H, W, device=DEV),
                                     torch.ones(1, 1, 3, 3, device=DEV), padding=1)
    assert torch.equal(out.float(), cnt.expand(1, Cout, H, W))
    g = torch.Generator(device="cpu").manual_seed(3)
    xr = torch.randint(-3, 4, (1, Cin, H, W), generator=g, dtype=torch.int8).to(DEV).contiguous(
        memory_format=torch.channels_last)
    wr = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g, dtype=torch.int8).to(DEV).contiguous(
        memory_format=torch.channels_last)
    wsr = wr.float().sum(dim=1, keepdim=True)
    o = C.qconv2d_w8_a8_ohalf(xr, wr, one, scal(1), scal(0), one, wsr, None, None, 1, 1)
    xs = torch.roll(xr, shifts=(5, 9), dims=(2, 3)).contiguous(memory_format=torch.channels_last)
    o2 = C.qconv2d_w8_a8_ohalf(xs, wr, one, scal(1), scal(0), one, wsr, None, None, 1, 1)
    assert torch.equal(torch.roll(o, shifts=(5, 9), dims=(2, 3))[:, :, 8:-8, 12:-12],
                       o2[:, :, 8:-8, 12:-12])
    ref = torch.nn.functional.conv2d(xr.float(), wr.float(), padding=1)
    assert torch.equal(o.float(), ref)   # random sums stay far below 2048: exact in f16

def test_qconv2d_errors(C):
    x = torch.zeros(1, 16, 4, 4, dtype=torch.int8, device=DEV)
    w = torch.zeros(8, 16, 3, 3, dtype=torch.int8, device=DEV)
    v = torch.zeros(8, device=DEV)
    ws = torch.zeros(8, 1, 3, 3, device=DEV)
    s0 = scal(0)
    with pytest.raises(RuntimeError, match="bias0 should equal output_channels"):
        C.qconv2d_w8_a8_ohalf(x, w, v, s0, s0, v, ws, None, None, 1, 0)
    with pytest.raises(RuntimeError, match="should equal K\\*R\\*S"):
        C.qconv2d_w8_a8_ohalf(x, w, v, s0, s0, v, None, v, None, 1, 1)
    with pytest.raises(RuntimeError, match="dilation must be 1"):
        C.qconv2d_w8_a8_ohalf(x, w, v, s0, s0, v, ws, None, None, 1, 1, 2)
    x6 = torch.zeros(1, 6, 4, 4, dtype=torch.int8, device=DEV)
    w6 = torch.zeros(8, 6, 3, 3, dtype=torch.int8, device=DEV)
    with pytest.raises(RuntimeError, match="alignment not to 4 is not supported"):
        C.qconv2d_w8_a8_ohalf(x6, w6, v, s0, s0, v, ws, None, None, 1, 1)


# ----------------------------------------------------------------------- fp16 debug GEMM (a10)
def test_qlinear_fp_reference(C, oracle):
    a = dd.f16(41, (64, 8), 0, 0.158)
    b = dd.f16(42, (8, 16), 0, 1.0)
    out = C.qlinear_fp_reference(t(a), t(b), t(dd.f16(43, (16,))))
    torch.testing.assert_close(out, torch.matmul(t(a), t(b)), rtol=1e-4, atol=1e-2)  # qlinear.py:95
    assert_bits_equal(out.cpu().numpy(), oracle.gemm_f16(a, b), "fp16 gemm")


# ------------------------------------------------------------------------------ graph capture
def test_ops_are_graph_capturable(C, oracle):
    """quantize_sdxl.py:184-286 captures the UNet in a CUDA graph; the ops must not sync or
    allocate outside torch's allocator.  Capture quantize + qlinear, replay on new data."""
    M, K, N = 64, 128, 64
    w = dd.int8(51, (N, K))
    b0, sc = dd.f32(52, (N,), -50, 50), dd.f32(53, (N,), 0.001, 0.01)
    wd, b0d, scd = t(w), t(b0), t(sc)
    x_static = t(dd.normal_f16(54, (M, K)))
    s_inv, zp = scal(20.0), scal(4.0)

    def fwd():
        q = C.quantize_per_tensor_to_int8(x_static, s_inv, zp)
        return C.qlinear_w8_a8_ohalf(q, wd, scd, s_inv, zp, b0d, scd, b0d, None)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_static = fwd()
    x2 = dd.normal_f16(55, (M, K))
    x_static.copy_(t(x2))
    g.replay()
    torch.cuda.synchronize()
    want = oracle.qlinear(oracle.quantize(x2, 20.0, 4.0, C.FLAGS & 1), w, b0, sc, None,
                          C.FLAGS & 1)
    assert_bits_equal(y_static.cpu().numpy(), want, "graph replay")


# -------------------------------------------------------------- residual folded into the epilogue
def test_qlinear_residual_epilogue(C, oracle):
    """D = f16(f16(epilogue) + residual): identical to the op followed by torch's half add."""
    M, K, N = 150, 128, 72
    a, w = dd.int8(61, (M, K)), dd.int8(62, (N, K))
    b0, sc = dd.f32(63, (N,), -50, 50), dd.f32(64, (N,), 0.001, 0.01)
    bias, res = dd.f16(65, (N,), -1, 1), dd.normal_f16(66, (M, N), 3.0)
    args = (t(a), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), t(bias))
    plain = C.qlinear_w8_a8_ohalf(*args)
    fused = C.qlinear_w8_a8_ohalf(*args, _residual=t(res))
    assert torch.equal(fused, plain + t(res))
    want = oracle.add_f16(oracle.qlinear(a, w, b0, sc, bias, C.FLAGS & 1), res)
    assert_bits_equal(fused.cpu().numpy(), want, "residual epilogue")
    for cfg in (1, 4, 35, 41, 45, 20):
        assert torch.equal(C.qlinear_w8_a8_ohalf(*args, _residual=t(res), _cfg=cfg), fused)


def test_qconv2d_residual_epilogue(C, oracle):
    case = next(c for c in CONV_CASES if c[0] == "conv_res_320")
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    xin, win = t(x).permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2)
    args = (xin, win, t(wscale), scal(in_scale), scal(in_zp), t(scale),
            t(wsum.reshape(k, 1, r, s)), None, t(bias), stride, pad)
    plain = C.qconv2d_w8_a8_ohalf(*args)
    res = t(dd.normal_f16(71, (n, h, w_, k), 2.0)).permute(0, 3, 1, 2)     # channels-last
    assert torch.equal(C.qconv2d_w8_a8_ohalf(*args, _residual=res), plain + res)
    per_img = t(dd.normal_f16(72, (n, k), 2.0))                            # time-embedding add
    assert torch.equal(C.qconv2d_w8_a8_ohalf(*args, _residual=per_img, _residual_per_image=True),
                       plain + per_img[:, :, None, None])


HALO_CASES = [  # n, h, w, c, k, bias, residual ("", "full", "image"), forced tile (0 = automatic)
    (2, 16, 16, 320, 320, True, "", 0),          # = conv_res_320's shape: C = 2.5 chunks, automatic
    (1, 16, 32, 128, 80, True, "full", 90),      # 8 x 16 patches, one whole chunk, conv2's residual
    (1, 16, 16, 64, 72, False, "", 91),          # half a chunk, N tail (72 < 80), 8 x 8 patches
    (2, 8, 16, 192, 168, True, "image", 90),     # 1.5 chunks, N tail in the third tile, temb add
    (1, 8, 8, 448, 84, True, "full", 91),        # a single patch: every pixel a border class; N % 8 == 4
    (1, 24, 48, 256, 160, False, "full", 0),     # several patches per row / column, two chunks
    (3, 32, 32, 640, 96, True, "image", 0),      # batch 3, five chunks
    (1, 16, 32, 320, 168, True, "full", 92),     # 16 x 16 patches, 64-byte chunks (5 of them), N tail
    (2, 32, 16, 64, 80, False, "image", 92),     # one chunk
    (1, 16, 16, 192, 72, True, "", 92),          # a single all-border patch
    (1, 32, 16, 320, 320, True, "full", 93),     # 16 x 16 patches x 160 channels on 4 x 2 waves, two channel tiles
    (2, 16, 32, 128, 168, True, "image", 93),    # N tail in the second channel tile (8 of 160)
    (1, 16, 16, 64, 72, False, "", 93),          # N tail inside the first wave group; all-border patch
    (8, 32, 32, 64, 160, True, "full", 0),       # automatic, batch 8 (too few 160-channel workgroups for 93)
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[f"n{c[0]}_{c[1]}x{c[2]}_c{c[3]}_k{c[4]}_{c[6] or 'plain'}_t{c[7]}"
                                                  for c in HALO_CASES])
def test_qconv2d_halo_kernel_bit_exact(C, oracle, case):
    """csrc/iconv.hip (3x3 / stride 1 / pad 1 with the input halo resident in LDS): the oracle's bits,
    and the implicit-GEMM family's, for partial channel chunks, N tails, patches that are all
    border, both residual forms and both patch shapes."""
    n, h, w_, c, k, has_bias, res_kind, tile = case
    x = dd.int8(901, (n, h, w_, c))
    wt = dd.int8(902, (k, 3, 3, c))
    scale = dd.f32(903, (k,), 1e-4, 6e-4)
    in_zp = -11.0
    bias = dd.f16(904, (k,), -1, 1) if has_bias else None
    wsum = wt.astype(np.float32).sum(axis=3, dtype=np.float32)
    assert C.conv_halo_select(n, h, w_, c, k, 3, 3, 1, 1) in (90, 91, 92, 93)
    args = (t(x).permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2), t(scale), scal(1.0), scal(in_zp),
            t(scale), t(wsum.reshape(k, 1, 3, 3)), None, None if bias is None else t(bias), 1, 1)
    kw = {}
    add = None
    if res_kind == "full":
        r = t(dd.normal_f16(905, (n, h, w_, k), 2.0)).permute(0, 3, 1, 2)
        kw, add = dict(_residual=r), r
    elif res_kind == "image":
        r = t(dd.normal_f16(906, (n, k), 2.0))
        kw, add = dict(_residual=r, _residual_per_image=True), r[:, :, None, None]
    got = C.qconv2d_w8_a8_ohalf(*args, _cfg=tile, **kw)
    want = torch.from_numpy(oracle.qconv2d(x, wt, scale, wsum, in_zp, None, bias, 1, 1, C.FLAGS & 1)
                            ).to(DEV).permute(0, 3, 1, 2)
    if add is not None:
        want = want + add                     # the epilogue add == a following torch half add
    assert torch.equal(got, want), f"halo kernel != oracle: {int((got != want).sum())} elements"
    assert torch.equal(got, C.qconv2d_w8_a8_ohalf(*args, _cfg=4, **kw)), "halo != implicit GEMM"


@pytest.mark.parametrize("n,h,w_,c,k,tile", [(2, 8, 8, 192, 168, 0), (1, 16, 8, 64, 80, 92), (1, 8, 8, 128, 168, 93), (1, 4, 8, 320, 72, 91),
                                             (3, 8, 16, 128, 96, 90)])
def test_qconv2d_upsample2x_reads_the_small_tensor(C, n, h, w_, c, k, tile):
    """MIXDQ_FLAG_UPSAMPLE2X: conv(nearest-2x-upsample(x)) from the [n, h, w] tensor == the conv on
    the materialised upsampling, bit for bit (Upsample2D; quantize commutes with nearest upsampling)."""
    x = t(dd.int8(921, (n, h, w_, c))).permute(0, 3, 1, 2)
    wt = dd.int8(922, (k, 3, 3, c))
    scale = t(dd.f32(923, (k,), 1e-4, 6e-4))
    bias = t(dd.f16(924, (k,), -1, 1))
    wsum = t(wt.astype(np.float32).sum(axis=3, dtype=np.float32).reshape(k, 1, 3, 3))
    args = (t(wt).permute(0, 3, 1, 2), scale, scal(1.0), scal(7.0), scale, wsum, None, bias, 1, 1)
    assert C.conv_upsample2x_supported(tuple(x.shape), (k, c, 3, 3), 1, 1)
    big = torch.nn.functional.interpolate(x.float(), scale_factor=2.0, mode="nearest").to(torch.int8
                                          ).contiguous(memory_format=torch.channels_last)
    want = C.qconv2d_w8_a8_ohalf(big, *args, _cfg=4)
    got = C.qconv2d_w8_a8_ohalf(x, *args, _cfg=tile, _upsample2x=True)
    assert got.shape == want.shape and torch.equal(got, want)
    with pytest.raises(RuntimeError, match="shape outside"):
        C.qconv2d_w8_a8_ohalf(x, *args, _cfg=4, _upsample2x=True)      # implicit-GEMM tile forced


def test_qconv2d_halo_kernel_range(C):
    """Outside its range the automatic choice is the implicit-GEMM family, and forcing it fails."""
    assert C.conv_halo_select(1, 12, 12, 960, 640, 3, 3, 1, 1) == 0      # H % 8 != 0
    assert C.conv_halo_select(1, 16, 16, 320, 320, 3, 3, 2, 1) == 0      # stride 2
    assert C.conv_halo_select(1, 16, 16, 320, 320, 1, 1, 1, 0) == 0      # 1x1
    assert C.conv_halo_select(1, 16, 16, 48, 320, 3, 3, 1, 1) == 0       # C % 64 != 0
    x = t(dd.int8(911, (1, 12, 12, 64))).permute(0, 3, 1, 2)
    w = t(dd.int8(912, (16, 3, 3, 64))).permute(0, 3, 1, 2)
    v = torch.ones(16, device=DEV)
    ws = torch.ones(16, 1, 3, 3, device=DEV)
    with pytest.raises(RuntimeError, match="shape outside"):
        C.qconv2d_w8_a8_ohalf(x, w, v, scal(1), scal(0), v, ws, None, None, 1, 1, _cfg=90)


@pytest.mark.parametrize("cfg", sorted(__import__("mixdq_amd._C", fromlist=["x"]).IGEMM_CONFIGS)
                         if torch.cuda.is_available() else [])
def test_every_kernel_configuration_is_bit_exact(C, oracle, cfg):
    """All tile / stage configurations of the igemm family give the oracle's bits (linear with a
    ragged M, K tail for BK = 128; padded conv with odd spatial size)."""
    M, K, N = 203, 1232, 136          # K % 128 != 0, N % 64 != 0
    a, w = dd.int8(81, (M, K)), dd.int8(82, (N, K))
    b0, sc = dd.f32(83, (N,), -500, 500), dd.f32(84, (N,), 1e-4, 1e-3)
    bias = dd.f16(85, (N,), -1, 1)
    out = C.qlinear_w8_a8_ohalf(t(a), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), t(bias),
                                _cfg=cfg)
    assert_bits_equal(out.cpu().numpy(), oracle.qlinear(a, w, b0, sc, bias, C.FLAGS & 1),
                      f"linear cfg {cfg}")
    # K % BK == 0: the Linear fast staging path (scalar K-tile advance, clamped tail rows), with
    # ragged M and N tails for every tile shape (64x80, 64x240, 128x320 ...) and > STAGES K-tiles
    M, K, N = 331, 1280, 424
    a, w = dd.int8(86, (M, K)), dd.int8(87, (N, K))
    b0, sc = dd.f32(88, (N,), -500, 500), dd.f32(89, (N,), 1e-4, 1e-3)
    out = C.qlinear_w8_a8_ohalf(t(a), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), None,
                                _cfg=cfg)
    assert_bits_equal(out.cpu().numpy(), oracle.qlinear(a, w, b0, sc, None, C.FLAGS & 1),
                      f"linear fast path cfg {cfg}")
    case = next(c for c in CONV_CASES if c[0] == "conv_s2_odd")
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    out = C.qconv2d_w8_a8_ohalf(t(x).permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2), t(wscale),
                                scal(in_scale), scal(in_zp), t(scale), t(wsum.reshape(k, 1, r, s)),
                                None, t(bias), stride, pad, _cfg=cfg)
    want = oracle.qconv2d(x, wt, scale, wsum, in_zp, None, bias, stride, pad, C.FLAGS & 1)
    assert_bits_equal(out.permute(0, 2, 3, 1).contiguous().cpu().numpy(), want, f"conv cfg {cfg}")


# --------------------------------------------------------------------- packed W4 weights (f-2)
W4_LINEAR = [(203, 1280, 136, True), (1024, 1280, 1280, False), (64, 5120, 640, True),
             (4096, 640, 640, True), (3000, 1280, 5120, False), (77, 2048, 640, False)]


@pytest.mark.parametrize("M,K,N,bias", W4_LINEAR)
def test_qlinear_w4_equals_w8_on_unpacked_values(C, oracle, M, K, N, bias):
    """MIXDQ_FLAG_W4: packed signed 4-bit weights, unpacked in the kernel.  Must equal the oracle's
    W8 restatement run on oracle.unpack_w4(packed) bit for bit (every kernel configuration)."""
    from mixdq_amd.nn.utils import pack_w4
    a = dd.int8(91, (M, K))
    q = dd.int8(92, (N, K), -8, 8)
    packed = pack_w4(torch.from_numpy(q))
    wsum = q.astype(np.float32).sum(axis=1, dtype=np.float32)
    b0 = (wsum * np.float32(-7.0)).astype(np.float32)
    sc = dd.f32(93, (N,), 1e-3, 1e-2)
    bs = dd.f16(94, (N,), -1, 1) if bias else None
    want = oracle.qlinear(a, oracle.unpack_w4(packed.numpy()), b0, sc, bs, C.FLAGS & 1)
    assert np.array_equal(oracle.unpack_w4(packed.numpy()), q)
    for cfg in (0,) + tuple(sorted(C.IGEMM_CONFIGS)):
        out = C.qlinear_w8_a8_ohalf(t(a), packed.to(DEV), t(sc), scal(1), scal(0), t(wsum), t(sc),
                                    t(b0), None if bs is None else t(bs), _w4=True, _cfg=cfg)
        assert_bits_equal(out.cpu().numpy(), want, f"w4 linear cfg {cfg}")


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[4] % 32 == 0 and c[0] in (
    "conv_res_320", "conv_res_960_640", "conv_down_s2", "conv_shortcut_1x1", "conv_odd_hw",
    "conv_tiny_hw", "conv_s2_odd")], ids=lambda c: c[0])
def test_qconv2d_w4_equals_w8_on_unpacked_values(C, oracle, case):
    from mixdq_amd.nn.utils import pack_w4
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x = dd.int8(seed + 1000, (n, h, w_, c))
    q = dd.int8(seed, (k, r, s, c), -8, 8)
    packed = pack_w4(torch.from_numpy(q))                          # [K, R, S, C/2]
    sc = dd.f32(seed + 2000, (k,), 1e-3, 1e-2)
    zp = np.float32(-11.0)
    bias = dd.f16(seed + 3000, (k,)) if has_bias else None
    wsum = q.astype(np.float32).sum(axis=3, dtype=np.float32)
    bias0 = (wsum.reshape(k, -1).sum(axis=1, dtype=np.float32) * zp).astype(np.float32)
    want = oracle.qconv2d(x, oracle.unpack_w4(packed.numpy()), sc, wsum if pad else None, zp,
                          bias0 if not pad else None, bias, stride, pad, C.FLAGS & 1)
    win = packed.to(DEV).permute(0, 3, 1, 2)                       # [K, C/2, R, S] channels-last
    for cfg in (0,) + tuple(sorted(C.IGEMM_CONFIGS)):
        out = C.qconv2d_w8_a8_ohalf(t(x).permute(0, 3, 1, 2), win, t(sc), scal(1), scal(zp), t(sc),
                                    t(wsum.reshape(k, 1, r, s)) if pad else None,
                                    t(bias0) if not pad else None,
                                    None if bias is None else t(bias), stride, pad, _w4=True,
                                    _cfg=cfg)
        assert_bits_equal(out.permute(0, 2, 3, 1).contiguous().cpu().numpy(), want,
                          f"w4 conv {name} cfg {cfg}")


def test_w4_rejects_unaligned_k(C):
    a = torch.zeros(4, 48, dtype=torch.int8, device=DEV)
    w = torch.zeros(8, 24, dtype=torch.int8, device=DEV)
    v = torch.zeros(8, device=DEV)
    with pytest.raises(RuntimeError, match="unsupported"):
        C.qlinear_w8_a8_ohalf(a, w, v, scal(0), scal(0), v, v, v, None, _w4=True)


# ------------------------------------------------------------------------- grouped launch (f-1)
@pytest.mark.parametrize("M,K,Ns,row_map,w4", [
    (76, 2048, [2560, 1280, 2560, 1280, 1280], (76, 77, 1), False),   # k|v projections, BOS rows
    (152, 2048, [640, 640], (76, 77, 1), False),                      # batch 2
    (1, 1280, [320, 640, 1280, 1280, 320], None, False),              # time-embedding projections
    (8, 1280, [320, 1280], None, False),
    (76, 2048, [1280, 2560], (76, 77, 1), True),                      # packed W4 members
    (200, 256, [136, 72, 8], None, False),                            # ragged N tails
])
def test_qlinear_grouped_members_equal_their_own_launches(C, oracle, M, K, Ns, row_map, w4):
    """mixdq_qlinear_w8a8_grouped: every member of the one grouped launch gets the oracle's bits
    (= what its own mixdq_qlinear_w8a8_rows launch gives), with per-member N, bias / no bias, the
    shared BOS row map, and rows outside the map left untouched."""
    from mixdq_amd.nn.utils import pack_w4
    a = dd.int8(101, (M, K))
    members, wants, outs = [], [], []
    for i, N in enumerate(Ns):
        q = dd.int8(110 + i, (N, K), -8 if w4 else -128, 8 if w4 else 128)
        b0, sc = dd.f32(120 + i, (N,), -300, 300), dd.f32(130 + i, (N,), 1e-4, 1e-3)
        bias = dd.f16(140 + i, (N,), -1, 1) if i % 2 else None
        wants.append(oracle.qlinear(a, q, b0, sc, bias, C.FLAGS & 1))
        wdev = pack_w4(torch.from_numpy(q)).to(DEV) if w4 else t(q)
        if row_map:
            g, stride, off = row_map
            out = torch.full((M // g, stride, N), 7.0, dtype=torch.float16, device=DEV)
        else:
            out = torch.full((M, N), 7.0, dtype=torch.float16, device=DEV)
        outs.append(out)
        members.append((wdev, t(b0), t(sc), None if bias is None else t(bias), out))
    table = C.GemmGroupTable(members, w4=w4)
    for cfg in (0, 37, 35, 41, 4, 56):
        for o in outs:
            o.fill_(7.0)
        C.qlinear_grouped(t(a), table, _row_map=row_map, _cfg=cfg)
        for i, (o, want) in enumerate(zip(outs, wants)):
            if row_map:
                g, stride, off = row_map
                got = o[:, off:off + g].reshape(M, -1)
                assert (o[:, :off] == 7.0).all()
            else:
                got = o
            assert_bits_equal(got.contiguous().cpu().numpy(), want, f"member {i} cfg {cfg}")
