"""CPU tests of the oracle itself: the C restatement against the committed golden vectors
(produced in the build container from the reference's own Python and torch's CPU quantizer), the
NumPy mirror, the reference's in-file formulas at the reference's tolerances, and the Path A
fake-quant restatement against outputs of the reference's QuantLayer."""
import hashlib

import numpy as np
import pytest
import torch

from tests import detdata as dd
from tests.cases import (LINEAR_CASES, CONV_CASES, MODULE_CASES, linear_inputs, conv_inputs,
                         make_float_module, module_input)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_half_converters_exhaustive(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([L.mixdq_oracle_h2f(int(h)) for h in allh], dtype=np.float32)
    nan = np.isnan(ref)
    assert (got.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all() and np.isnan(got[nan]).all()
    # every finite half round-trips; f32 -> f16 rounding against NumPy on random + edge floats
    fin = allh[~nan & ~np.isinf(ref)]
    back = np.array([L.mixdq_oracle_f2h(float(v)) for v in ref[~nan & ~np.isinf(ref)]], np.uint16)
    assert (back == fin).all()
    x = (dd.uniform01(5, (20000,)) * 2 - 1).astype(np.float32) * np.float32(10.0) ** (
        dd.int8(6, (20000,), -8, 6).astype(np.float32))
    x = np.concatenate([x, np.array([65504, 65519.996, 65520, 65536, 1e10, -1e10, 2.0 ** -25,
                                     2.0 ** -25 * 1.0001, 2.0 ** -24, 3e-8, 6.1e-5, 0, -0.0],
                                    dtype=np.float32)])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([L.mixdq_oracle_f2h(float(v)) for v in x], dtype=np.uint16)
    assert (got == want).all()


def test_quantize_matches_torch_quantize_per_tensor(oracle, ops_golden, ops_small):
    """The reference's own exact check (op/quant.py:24-27), on CPU, live and against the fixture."""
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_ref")
    x = dd.f16(case["seed"], tuple(case["shape"]))
    t = torch.from_numpy(x).float()
    zp = torch.round((t.max() + t.min()) / 2)
    scale = (t.max() - t.min()) / 255
    ref = torch.quantize_per_tensor(t, scale, zp, torch.qint8).int_repr().numpy()
    assert np.array_equal(ref, ops_small[case["expect"]])
    for variant in (0, 1):
        assert np.array_equal(oracle.quantize(x, case["scale_inv"], case["zp"], variant), ref)
        assert np.array_equal(oracle.np_quantize(x, case["scale_inv"], case["zp"], variant), ref)


def test_quantize_golden_hashes_and_edges(oracle, ops_golden, ops_small):
    for case in ops_golden["quantize"]:
        nm = case["name"]
        if nm.startswith("q_act"):
            x = dd.normal_f16(case["seed"], tuple(case["shape"]), std=case["std"])
            assert sha(oracle.quantize(x, case["scale_inv"], case["zp"], 0)) == case["sha_A"]
            assert sha(oracle.quantize(x, case["scale_inv"], case["zp"], 1)) == case["sha_B"]
        elif nm.startswith("q_edge"):
            got = oracle.quantize(ops_small[case["x"]], case["scale_inv"], case["zp"], 0)
            assert np.array_equal(got, ops_small[case["expect"]])
        elif nm == "q_sep":
            x, si, zp = ops_small[case["x"]], ops_small[case["scale_inv"]], ops_small[case["zp"]]
            assert x.size == case["n"] >= 16
            for i in range(x.size):
                a = oracle.quantize(x[i:i + 1], float(si[i]), float(zp[i]), 0)[0]
                b = oracle.quantize(x[i:i + 1], float(si[i]), float(zp[i]), 1)[0]
                assert a == ops_small[case["expect_A"]][i] and b == ops_small[case["expect_B"]][i]
                assert a != b    # the separators really separate the two rounding variants
        elif nm == "q_allhalf":
            allh = np.arange(1, 0x7c00, dtype=np.uint16).view(np.float16)
            allh = np.concatenate([allh, -allh])
            assert sha(oracle.quantize(allh, case["scale_inv"], case["zp"], 0)) == case["sha_A"]
            assert sha(oracle.quantize(allh, case["scale_inv"], case["zp"], 1)) == case["sha_B"]


def test_quantize_strided_semantics(oracle, ops_golden, ops_small):
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_bos_slice")
    x = dd.normal_f16(case["seed"], tuple(case["shape"]))
    got = oracle.quantize(x[:, 1:, :], case["scale_inv"], case["zp"])
    assert np.array_equal(got, ops_small[case["expect"]])
    assert np.array_equal(got, oracle.quantize(np.ascontiguousarray(x[:, 1:, :]),
                                               case["scale_inv"], case["zp"]))


@pytest.mark.parametrize("case", LINEAR_CASES, ids=[c[0] for c in LINEAR_CASES])
def test_qlinear_golden(oracle, ops_golden, case):
    g = next(c for c in ops_golden["qlinear"] if c["name"] == case[0])
    a, w, wscale, in_scale, in_zp, bias, scale, bias0 = linear_inputs(case)
    D, acc = oracle.qlinear(a, w, bias0, scale, bias, 0, return_acc=True)
    assert sha(acc) == g["sha_acc"] and sha(D) == g["sha_A"]
    assert sha(oracle.qlinear(a, w, bias0, scale, bias, 1)) == g["sha_B"]
    flat = D.reshape(-1).view(np.uint16)
    assert flat[np.asarray(g["sample_idx"])].tolist() == g["sample_bits_A"]
    # independent NumPy restatement agrees bit for bit
    assert np.array_equal(oracle.np_qlinear(a, w, bias0, scale, bias, 0).view(np.uint16),
                          D.view(np.uint16))
    assert np.array_equal(acc.reshape(-1, g["N"]), a.astype(np.int32) @ w.astype(np.int32).T)


def test_qlinear_reference_formulas(oracle):
    """op/qlinear.py:66-101: the reference's integer and FP references at its tolerances."""
    case = LINEAR_CASES[0]
    a, w, wscale, in_scale, in_zp, bias, scale, bias0 = linear_inputs(case)
    out = torch.from_numpy(oracle.qlinear(a, w, bias0, scale, bias))
    ai, wi, ws, bs = map(torch.from_numpy, (a, w, wscale, bias))
    infused = ws * float(in_scale)
    offset = ws * wi.to(torch.int32).sum(dim=1) * (float(in_zp) * float(in_scale))
    ref_int = (torch.matmul(ai.float(), wi.float().t()) * infused - offset + bs.float()).half()
    ref_fp = (torch.matmul((ai.float() - float(in_zp)) * float(in_scale),
                           (wi.float() * ws[:, None]).t()) + bs.float()).half()
    torch.testing.assert_close(out, ref_int, atol=1e-4, rtol=1e-2)
    torch.testing.assert_close(out, ref_fp, atol=1e-2, rtol=1e-2)


SMALL_CONV = [c for c in CONV_CASES if c[4] * c[5] * c[6] * c[7] * c[1] * c[2] * c[3] < 3e8]


@pytest.mark.parametrize("case", SMALL_CONV, ids=[c[0] for c in SMALL_CONV])
def test_qconv2d_golden(oracle, ops_golden, case):
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    g = next(cc for cc in ops_golden["qconv2d"] if cc["name"] == name)
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    D, acc = oracle.qconv2d(x, wt, scale, wsum if pad > 0 else None, in_zp,
                            bias0 if pad == 0 else None, bias, stride, pad, 0, return_acc=True)
    assert sha(acc) == g["sha_acc"] and sha(D) == g["sha_A"]
    # exact integer accumulators against torch's float conv (|acc| < 2^24 for these sizes)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(),
                                     torch.from_numpy(wt).permute(0, 3, 1, 2).double(),
                                     stride=stride, padding=pad).permute(0, 2, 3, 1)
    assert np.array_equal(acc, ref.numpy().astype(np.int32))
    if pad > 0:   # materialised zero-point propagation == fused form
        b0 = oracle.zp_propagate(wsum, in_zp, n, h, w_, stride, pad)
        D2 = oracle.np_epilogue(acc, b0, scale[None, None, None, :],
                                None if bias is None else bias[None, None, None, :], 0)
        assert np.array_equal(D2.view(np.uint16), D.view(np.uint16))


def test_qconv2d_reference_formula(oracle):
    """op/qconv2d.py:65-101: kernel vs the reference's integer reference with default fp16
    assert_close tolerances (rtol 1e-3, atol 1e-5)."""
    case = next(c for c in CONV_CASES if c[0] == "conv_ref_07")
    name, n, h, w_, c, k, r, s, pad, stride, has_bias, rng, seed = case
    x, wt, wscale, in_scale, in_zp, bias, scale, wsum, bias0 = conv_inputs(case)
    out = torch.from_numpy(oracle.qconv2d(x, wt, scale, wsum, in_zp, None, bias, stride, pad))
    xi = torch.from_numpy(x).permute(0, 3, 1, 2).float()
    wi = torch.from_numpy(wt).permute(0, 3, 1, 2).float()
    accf = torch.nn.functional.conv2d(xi, wi, stride=stride, padding=pad)
    a_ = torch.broadcast_to(torch.tensor(-1.0 * in_zp), (n, 1, h, w_))
    b0 = torch.nn.functional.conv2d(a_, wi.sum(dim=1, keepdim=True), stride=stride, padding=pad)
    ref = (accf + b0) * torch.from_numpy(wscale)[None, :, None, None] * float(in_scale)
    ref = (ref + torch.from_numpy(bias)[None, :, None, None]).half().permute(0, 2, 3, 1)
    torch.testing.assert_close(out, ref, rtol=1e-3, atol=1e-5)


def test_add_f16_matches_torch(oracle):
    a, b = dd.normal_f16(1, (4096,), 100.0), dd.normal_f16(2, (4096,), 3.0)
    want = (torch.from_numpy(a) + torch.from_numpy(b)).numpy()
    assert np.array_equal(oracle.add_f16(a, b).view(np.uint16), want.view(np.uint16))


@pytest.mark.parametrize("c", MODULE_CASES, ids=[c["key"] for c in MODULE_CASES])
@pytest.mark.parametrize("w_bits", [8, 4])
def test_fakequant_restatement_matches_reference_quantlayer(modules_golden, fakequant_golden, c,
                                                            w_bits):
    """oracle/fakequant.py == the reference's QuantLayer (Path A) outputs stored in
    fakequant.npz, bit for bit (same torch ops on the same CPU tensors, fp32)."""
    from oracle.fakequant import quant_layer_forward
    key = c["key"]
    fm = make_float_module(c)
    x = module_input(c).float()
    idx = {2: 0, 4: 1, 8: 2}

    def ck(sfx, field):   # un-rounded quantizer state is not stored; the fixture holds the fp16
        return torch.from_numpy(modules_golden[f"{key}.ckpt.{sfx}.{field}"]).float()

    # The QuantLayer ran with fp32 deltas; fakequant.npz was produced with those.  Recompute the
    # fp32 deltas exactly as init_quant_params does (min-max), then compare.
    split = c.get("split", 0)

    from mixdq_amd.calib import ActRange, weight_delta as wdelta

    def aparams(xx):   # one calibration forward, 8-bit statistics (index 2 of [2, 4, 8])
        r = ActRange()
        r.update(xx)
        return r.params(2)

    kw = None
    if c["kind"] == "conv":
        kw = dict(stride=fm.stride, padding=fm.padding, dilation=fm.dilation, groups=fm.groups)
    torch.set_grad_enabled(False)   # the fixture was produced under no_grad (same conv backend)
    if split:
        ad, az = aparams(x[:, :split])
        ad0, az0 = aparams(x[:, split:])
        y = quant_layer_forward(x, fm.weight, fm.bias, wdelta(fm.weight[:, :split], w_bits), ad, az,
                                w_bits, 8, kw, split, wdelta(fm.weight[:, split:], w_bits), ad0, az0)
    else:
        ad, az = aparams(x)
        y = quant_layer_forward(x, fm.weight, fm.bias, wdelta(fm.weight, w_bits), ad, az, w_bits, 8,
                                kw)
    torch.set_grad_enabled(True)
    want = fakequant_golden[f"{key}.pathA_w{w_bits}a8"]
    assert np.array_equal(y.detach().numpy(), want)
    # and the fp16 checkpoint of the fixture is that state, rounded (convert_ckpt.py:36): pins
    # mixdq_amd.calib against the reference's quantizer initialisation
    from mixdq_amd.calib import weight_quantizer_entry
    w_first = fm.weight[:, :split] if split else fm.weight
    assert torch.equal(ck("weight_quantizer", "delta_list"),
                       weight_quantizer_entry(w_first)["delta_list"].float())
    r = ActRange()
    r.update(x[:, :split] if split else x)
    assert torch.equal(ck("act_quantizer", "delta_list"), r.entry()["delta_list"].float())
    assert torch.equal(ck("act_quantizer", "zero_point_list"),
                       r.entry()["zero_point_list"].float())


# ----------------------------------------------------------- producer fusions / attention (f-1)
def _ulp16(ref: torch.Tensor) -> torch.Tensor:
    a = ref.float().abs().clamp(min=2.0 ** -14)
    return 2.0 ** (torch.floor(torch.log2(a)) - 10)


def test_layernorm_restatement_matches_torch_fp32_reference(oracle):
    """oracle.layernorm_quantize (the kernels' bit-level specification) against PyTorch's FP32
    LayerNorm rounded to FP16: within one FP16 ulp; each INT8 output is quantize() of that FP16."""
    x = dd.normal_f16(301, (37, 640), 1.7)
    gamma = (dd.normal_f16(302, (640,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(303, (640,), 0.2)
    qp = [(25.0, -3.0), (40.0, 11.0)]
    outs, h = oracle.layernorm_quantize(x, gamma, beta, 1e-5, qp)
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x).float(), (640,),
                                         torch.from_numpy(gamma).float(),
                                         torch.from_numpy(beta).float(), 1e-5)
    got = torch.from_numpy(h).float()
    assert bool(((got - ref).abs() <= 1.001 * _ulp16(ref)).all())
    for (si, zp), q in zip(qp, outs):
        assert np.array_equal(q, oracle.quantize(h, si, zp))


@pytest.mark.parametrize("C", [64, 96, 128, 640, 1280, 2048])
def test_layernorm_reduction_order_is_stable_for_shifted_rows(oracle, C):
    """The row statistics are per-16-column (sum, centred sum of squares) partials folded in a fixed order
    (oracle/mixdq_oracle.c: the order the GEMM epilogue of csrc/igemm_ln.hip can follow from its column tiles)
    and combined with Chan's formula, so the variance does not suffer E[x^2] - mean^2 cancellation: rows whose
    mean is 12 standard deviations away from zero (mean^2 / var = 144) stay within one FP16 ulp (+ the FP32
    noise of the mean itself, 4e-6) of PyTorch's FP64 LayerNorm of the same FP16 input.  Widths cover every
    segment count (C / 16 = 4, 6, 8, 40, 80, 128)."""
    x = (dd.normal_f16(305, (19, C), 0.5).astype(np.float32) + 6.0).astype(np.float16)
    gamma = (dd.normal_f16(307, (C,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(308, (C,), 0.2)
    _, h = oracle.layernorm_quantize(x, gamma, beta, 1e-5, [])
    ref = torch.nn.functional.layer_norm(torch.from_numpy(x).double(), (C,), torch.from_numpy(gamma).double(),
                                         torch.from_numpy(beta).double(), 1e-5).float()
    got = torch.from_numpy(h).float()
    assert bool(((got - ref).abs() <= 1.001 * _ulp16(ref) + 4e-6).all())


def test_groupnorm_silu_restatement_matches_torch_fp32_reference(oracle):
    x = (dd.normal_f16(311, (2, 6, 5, 64), 1.5).astype(np.float32) +
         dd.normal_f16(312, (1, 1, 1, 64), 0.7).astype(np.float32)).astype(np.float16)   # NHWC
    gamma = (dd.normal_f16(313, (64,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(314, (64,), 0.2)
    q, h = oracle.groupnorm_silu_quantize(x, gamma, beta, 1e-5, 8, False, 30.0, -20.0)
    xt = torch.from_numpy(x).float().permute(0, 3, 1, 2)
    ref = torch.nn.functional.group_norm(xt, 8, torch.from_numpy(gamma).float(),
                                         torch.from_numpy(beta).float(), 1e-5).permute(0, 2, 3, 1)
    got = torch.from_numpy(h).float()
    assert bool(((got - ref).abs() <= 1.001 * _ulp16(ref) + 1e-6).all())
    assert np.array_equal(q, oracle.quantize(h, 30.0, -20.0))
    # + SiLU: SiLU of the restatement's own FP16 normalised value, one more rounding
    _, hs = oracle.groupnorm_silu_quantize(x, gamma, beta, 1e-5, 8, True, 30.0, -20.0)
    ref_s = torch.nn.functional.silu(got)
    assert bool(((torch.from_numpy(hs).float() - ref_s).abs() <= 1.001 * _ulp16(ref_s) + 1e-7).all())


def test_geglu_restatement_matches_torch_fp32_reference(oracle):
    hin = dd.normal_f16(321, (19, 2 * 96), 2.0)
    q, o = oracle.geglu_quantize(hin, 20.0, -100.0)
    t = torch.from_numpy(hin).float()
    ref = (t[:, :96] * torch.nn.functional.gelu(t[:, 96:]).half().float()).half().float()
    got = torch.from_numpy(o).float()
    atol = 4e-7 * t[:, :96].abs() * t[:, 96:].abs().clamp(min=1.0)     # erf cancellation, gate << 0
    assert bool(((got - ref).abs() <= 2.001 * _ulp16(ref) + atol).all())
    assert np.array_equal(q, oracle.quantize(o, 20.0, -100.0))


def test_attention_restatement_matches_torch(oracle):
    q = dd.normal_f16(331, (2, 50, 128), 1.2)
    k = dd.normal_f16(332, (2, 77, 128), 1.2)
    v = dd.normal_f16(333, (2, 77, 128), 1.2)
    o16, o64 = oracle.attention_f16(q, k, v, 2)

    def heads(a):
        return torch.from_numpy(a).double().unflatten(-1, (2, 64)).transpose(1, 2)
    ref = torch.nn.functional.scaled_dot_product_attention(heads(q), heads(k), heads(v))
    ref = ref.transpose(1, 2).reshape(2, 50, 128).numpy()
    assert np.abs(o64 - ref).max() < 1e-12
    assert np.array_equal(o16.view(np.uint16), ref.astype(np.float16).view(np.uint16))
