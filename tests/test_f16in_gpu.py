"""GPU parity of the quantize-in-prologue GEMM (mixdq_qlinear_f16in_w8a8, csrc/igemm_aq.hip): ONE launch
must give the bits of the reference's pair  quant_op(x) -> qlinear / 1x1 qconv2d  (nn/Linear.py:162-176,
nn/Conv2d.py:294-311) -- checked against the oracle's quantize -> qlinear restatements (bit-exact, both
rounding variants), for every tile configuration of the family, with M / N tails, a K as short as one
K-tile and as long as 40, the separators of the two quantize roundings, saturating / non-finite inputs,
the BOS row map, column slices (split shortcut halves), residual epilogue and packed-W4 weights."""
import numpy as np
import pytest
import torch

from tests import detdata as dd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["A", "B"], ids=["fma", "mul_add"])
def epilogue_variant(request, monkeypatch):
    import mixdq_amd._C as C_
    monkeypatch.setattr(C_, "FLAGS", 1 if request.param == "B" else 0)
    return request.param


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def scal(v):
    return torch.tensor(float(v), dtype=torch.float32, device=DEV)


def bits_equal(got, want, what):
    g = got.view(np.uint16) if got.dtype == np.float16 else got
    w = want.view(np.uint16) if want.dtype == np.float16 else want
    assert g.shape == w.shape, f"{what}: shape {g.shape} vs {w.shape}"
    bad = np.nonzero(g.reshape(-1) != w.reshape(-1))[0]
    assert bad.size == 0, (f"{what}: {bad.size}/{g.size} differ; first at {bad[0]}: "
                           f"got {got.reshape(-1)[bad[0]]!r} want {want.reshape(-1)[bad[0]]!r}")


def problem(seed, M, K, N, bias=True, lo=-4.0, hi=4.0):
    x = dd.f16(seed, (M, K), lo, hi)
    w = dd.int8(seed + 1, (N, K))
    sc = dd.f32(seed + 2, (N,), 1e-4, 1e-3)
    b0 = dd.f32(seed + 3, (N,), -500, 500)
    bs = dd.f16(seed + 4, (N,), -1, 1) if bias else None
    return x, w, sc, b0, bs


def want_of(oracle, x, s_inv, zp, w, b0, sc, bs, variant):
    q = oracle.quantize(x, s_inv, zp, variant)
    return oracle.qlinear(q.reshape(-1, x.shape[-1]), w, b0, sc, bs, variant)


CFGS = [4, 13, 27, 28, 35, 37, 41, 44, 45, 56]


@pytest.mark.parametrize("cfg", [0] + CFGS)
def test_f16in_every_configuration_is_bit_exact(C, oracle, cfg):
    """Ragged M and N tails on every tile, more K-tiles than stages; scale / zero point that saturate
    part of the input."""
    assert tuple(C.F16IN_CONFIGS) == tuple(CFGS)
    M, K, N = 331, 1280, 424
    x, w, sc, b0, bs = problem(300, M, K, N)
    s_inv, zp = 31.37, -9.0                      # |x| <= 4 -> |x * s_inv| up to 125: both rails clamp
    out = C.qlinear_f16in(t(x), scal(s_inv), scal(zp), t(w), t(sc), t(b0), t(bs), _cfg=cfg)
    assert out.dtype == torch.float16 and tuple(out.shape) == (M, N)
    bits_equal(out.cpu().numpy(), want_of(oracle, x, s_inv, zp, w, b0, sc, bs, C.FLAGS & 1), f"cfg {cfg}")


@pytest.mark.parametrize("M,K,N", [(1024, 1280, 1280), (1024, 5120, 1280), (1024, 1280, 10240),
                                   (4096, 640, 640), (4096, 2560, 640), (4096, 640, 5120),
                                   (76, 2048, 1280), (1, 1280, 1280), (2, 2816, 1280), (64, 128, 64),
                                   (1000, 640, 320), (16384, 640, 320)])
def test_f16in_unet_shapes_automatic_choice(C, oracle, M, K, N):
    x, w, sc, b0, bs = problem(310 + M % 97, M, K, N, bias=(N % 3 != 0))
    s_inv, zp = 17.3, 3.0
    assert C.qlinear_f16in_supported(t(x), N, K)
    out = C.qlinear_f16in(t(x), scal(s_inv), scal(zp), t(w), t(sc), t(b0), None if bs is None else t(bs))
    bits_equal(out.cpu().numpy(), want_of(oracle, x, s_inv, zp, w, b0, sc, bs, C.FLAGS & 1), f"{M}x{N}x{K}")


def test_f16in_equals_the_two_launches_on_hard_inputs(C, oracle, ops_golden, ops_small):
    """Every finite half, +-inf, NaN, -0, the FMA / mul+add separators of the quantize fixture: the fused
    launch must give exactly what quantize_per_tensor_to_int8 -> qlinear_w8_a8_ohalf give."""
    K, N = 256, 64
    allh = np.arange(65536, dtype=np.uint16).view(np.float16)          # 256 rows of 256: every bit pattern
    x = allh.reshape(256, K)
    w = dd.int8(321, (N, K))
    sc, b0 = dd.f32(322, (N,), 1e-4, 1e-3), dd.f32(323, (N,), -50, 50)
    for s_inv, zp in [(0.0312, -3.0), (8.0, 0.0), (123.456, 17.0), (1e-3, -128.0)]:
        q = C.quantize_per_tensor_to_int8(t(x), scal(s_inv), scal(zp))
        pair = C.qlinear_w8_a8_ohalf(q, t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), None)
        one = C.qlinear_f16in(t(x), scal(s_inv), scal(zp), t(w), t(sc), t(b0), None)
        assert torch.equal(one.view(torch.int16), pair.view(torch.int16)), (s_inv, zp)
        bits_equal(q.cpu().numpy(), oracle.quantize(x, s_inv, zp, C.FLAGS & 1), "quantize")
    # the fixture's (x, s_inv, zp) triples on which fma(x, s, zp) and (x * s) + zp round differently: the
    # launch must follow the selected variant.  One-hot weights, scale 1, bias0 0: out[0, n] = f16(q[0, n])
    case = next(c for c in ops_golden["quantize"] if c["name"] == "q_sep")
    xs, si, zps = ops_small[case["x"]], ops_small[case["scale_inv"]], ops_small[case["zp"]]
    want = ops_small[case["expect_B" if C.FLAGS & 1 else "expect_A"]]
    eye = torch.eye(128, dtype=torch.int8, device=DEV)
    one, zero = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
    for i in range(xs.size):
        row = t(np.full((1, 128), xs[i], dtype=np.float16))
        out = C.qlinear_f16in(row, scal(si[i]), scal(zps[i]), eye, one, zero, None)
        assert (out == float(want[i])).all(), (i, xs[i], si[i], zps[i], out[0, :4], want[i])


def test_f16in_bos_rows_read_and_written_in_place(C, oracle):
    """QuantizedLinear's BOS path at batch 3: tokens 1.. of a [B, T, K] tensor are the operand (row map on
    the input), rows 1.. of [B, T, N] the output; token 0 of both is never touched."""
    B, T, K, N = 3, 77, 2048, 640
    x = dd.f16(330, (B, T, K), -3, 3)
    x[:, 0, :] = np.float16(np.nan)                # reading token 0 would poison the output
    w = dd.int8(331, (N, K))
    sc, b0 = dd.f32(332, (N,), 1e-4, 1e-3), dd.f32(333, (N,), -50, 50)
    out = torch.full((B, T, N), 7.0, dtype=torch.float16, device=DEV)
    assert C.qlinear_f16in_supported(t(x), N, K, bos=True)
    C.qlinear_f16in(t(x), scal(20.0), scal(-5.0), t(w), t(sc), t(b0), None, _out=out, _bos=True)
    want = want_of(oracle, np.ascontiguousarray(x[:, 1:, :]), 20.0, -5.0, w, b0, sc, None, C.FLAGS & 1)
    bits_equal(out[:, 1:].reshape(-1, N).cpu().numpy(), want, "bos body")
    assert (out[:, 0] == 7.0).all()


def test_f16in_column_slices_and_residual(C, oracle):
    """The split shortcut: x[..., :s] and x[..., s:] of one NHWC tensor read in place (row stride = all
    channels), the second launch adds the first's output after its own FP16 rounding."""
    M, Ca, Cb, N = 1024, 1280, 640, 1280
    xall = dd.f16(340, (M, Ca + Cb), -2, 2)
    xa, xb = xall[:, :Ca], xall[:, Ca:]
    wa, wb = dd.int8(341, (N, Ca)), dd.int8(342, (N, Cb))
    sc, b0 = dd.f32(343, (N,), 1e-4, 1e-3), dd.f32(344, (N,), -50, 50)
    bs = dd.f16(345, (N,), -1, 1)
    xd = t(xall)
    v = C.FLAGS & 1
    first = C.qlinear_f16in(xd[:, :Ca], scal(30.0), scal(1.0), t(wa), t(sc), t(b0), t(bs))
    want1 = want_of(oracle, np.ascontiguousarray(xa), 30.0, 1.0, wa, b0, sc, bs, v)
    bits_equal(first.cpu().numpy(), want1, "first half")
    second = C.qlinear_f16in(xd[:, Ca:], scal(25.0), scal(-2.0), t(wb), t(sc), t(b0), None, _residual=first)
    want2 = oracle.add_f16(want_of(oracle, np.ascontiguousarray(xb), 25.0, -2.0, wb, b0, sc, None, v), want1)
    bits_equal(second.cpu().numpy(), want2, "second half + residual")


def test_f16in_w4_weights(C, oracle):
    from mixdq_amd.nn.utils import pack_w4
    M, K, N = 203, 1280, 136
    x = dd.f16(350, (M, K), -4, 4)
    q = dd.int8(351, (N, K), -8, 8)
    packed = pack_w4(torch.from_numpy(q))
    sc, b0 = dd.f32(352, (N,), 1e-3, 1e-2), dd.f32(353, (N,), -50, 50)
    want = want_of(oracle, x, 30.0, 2.0, oracle.unpack_w4(packed.numpy()), b0, sc, None, C.FLAGS & 1)
    for cfg in [0] + CFGS:
        out = C.qlinear_f16in(t(x), scal(30.0), scal(2.0), packed.to(DEV), t(sc), t(b0), None, _w4=True,
                              _cfg=cfg)
        bits_equal(out.cpu().numpy(), want, f"w4 cfg {cfg}")


def test_f16in_range_and_errors(C):
    x = torch.zeros((64, 200), dtype=torch.float16, device=DEV)        # K % 64 != 0: outside the family
    assert not C.qlinear_f16in_supported(x, 64, 200)
    w = torch.zeros((64, 200), dtype=torch.int8, device=DEV)
    v = torch.ones(64, device=DEV)
    with pytest.raises(RuntimeError, match="shape outside"):
        C.qlinear_f16in(x, scal(1), scal(0), w, v, v, None)
    assert not C.qlinear_f16in_supported(x.float(), 64, 200)
    xt = torch.zeros((256, 64), dtype=torch.float16, device=DEV).t()   # last dimension not contiguous
    assert not C.qlinear_f16in_supported(xt, 64, 256)
    with pytest.raises(RuntimeError, match="fp16"):
        C.qlinear_f16in(x.float(), scal(1), scal(0), w, v, v, None)


def test_f16in_graph_capture_and_batch_rows(C, oracle):
    """Capturable (no host sync, scalars read on the device), and row i of a batched launch equals the
    launch on row block i alone (tile choice may differ: integer accumulation is exact)."""
    K, N = 1280, 1280
    x = dd.f16(360, (2048, K), -3, 3)
    w = dd.int8(361, (N, K))
    sc, b0 = dd.f32(362, (N,), 1e-4, 1e-3), dd.f32(363, (N,), -50, 50)
    xd, wd, scd, b0d, si, zp = t(x), t(w), t(sc), t(b0), scal(21.0), scal(4.0)
    ref = C.qlinear_f16in(xd, si, zp, wd, scd, b0d, None)
    half = C.qlinear_f16in(xd[1024:], si, zp, wd, scd, b0d, None)
    assert torch.equal(ref[1024:].view(torch.int16), half.view(torch.int16))
    out = torch.empty_like(ref)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        C.qlinear_f16in(xd, si, zp, wd, scd, b0d, None, _out=out)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    out.zero_()
    with torch.cuda.graph(g):
        C.qlinear_f16in(xd, si, zp, wd, scd, b0d, None, _out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


def test_modules_take_the_one_launch_path_and_keep_their_bits(C, monkeypatch):
    """QuantizedLinear (dense and BOS at batch 2) and the split 1x1 QuantizedConv2d at UNet sizes: forward
    under MIXDQ_F16IN=1 launches no quantize kernel for the supported shapes and returns the bits of
    MIXDQ_F16IN=0 (the reference's flow: nn/Linear.py:162-194, nn/Conv2d.py:294-347)."""
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from tests.cases import make_float_module
    from tests.test_host import prepared

    def synth(c):
        """A module of case `c` over a synthetic kernel-format checkpoint (convert_ckpt.py layout)."""
        name, oc = c["name"], c["cout"]
        fm = make_float_module(c)
        w = fm.weight.detach().float()
        golden = {f"{c['key']}.bos_pre_computed": dd.f16(c["seed"] + 7, (1, 1, oc), -1, 1)} if c.get("bos") else {}
        ck = {}
        halves = [("", slice(None))] if not c.get("split") else [("", slice(0, c["split"])), ("_0", slice(c["split"], None))]
        for i, (sfx, sl) in enumerate(halves):
            d = (w[:, sl].reshape(oc, -1).abs().amax(dim=1) / 127).half()
            ck[f"{name}.weight_quantizer{sfx}"] = {"delta_list": d[None].repeat(3, 1),
                                                   "zero_point_list": torch.zeros(3, oc).half()}
            ck[f"{name}.act_quantizer{sfx}"] = {"delta_list": torch.tensor([0.05 + 0.01 * i] * 3).half(),
                                                "zero_point_list": torch.tensor([120.0 + 9 * i] * 3).half()}
        cls = QuantizedLinear if c["kind"] == "linear" else QuantizedConv2d
        return cls.from_float(prepared(c, golden), split=c.get("split", 0), ckpt=ck).to(DEV)

    def make_qlinear(cin, cout, seed, bos=False):
        blk = "down_blocks.2.attentions.0.transformer_blocks.0."
        return synth(dict(key=f"l{seed}", kind="linear", name=blk + ("attn2.to_k" if bos else "attn1.to_q"),
                          cin=cin, cout=cout, bias=not bos, seed=seed, bos=bos))

    def make_qconv(cin, cout, ksize, seed, split=0):
        return synth(dict(key=f"c{seed}", kind="conv", name="up_blocks.0.resnets.2.conv_shortcut", cin=cin,
                          cout=cout, ksize=ksize, stride=1, pad=0, bias=True, seed=seed, split=split))
    calls = []
    real_q = C.quantize_per_tensor_to_int8_vectorized
    import mixdq_amd.nn.Linear as L_
    import mixdq_amd.nn.Conv2d as C2_

    def spy(*a, **k):
        calls.append(1)
        return real_q(*a, **k)
    monkeypatch.setattr(L_, "quant_op", spy)
    monkeypatch.setattr(C2_, "quant_op", spy)
    lin = make_qlinear(1280, 1280, seed=371)
    bos = make_qlinear(2048, 1280, seed=372, bos=True)
    conv = make_qconv(1920, 1280, 1, seed=373, split=1280)
    x = torch.from_numpy(dd.f16(374, (2, 1024, 1280), -3, 3)).to(DEV)
    xb = torch.from_numpy(dd.f16(375, (2, 77, 2048), -3, 3)).to(DEV)
    xc = torch.from_numpy(dd.f16(376, (2, 1920, 32, 32), -3, 3)).to(DEV).contiguous(
        memory_format=torch.channels_last)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setattr(C, "F16IN", mode)
        del calls[:]
        with torch.no_grad():
            outs[mode] = [lin(x), bos(xb), conv(xc)]
        assert len(calls) == (4 if mode == "0" else 0), (mode, len(calls))
    for a, b in zip(outs["0"], outs["1"]):
        assert a.shape == b.shape and torch.equal(a.contiguous().view(torch.int16), b.contiguous().view(torch.int16))


def test_f16in_conv_traces_its_persistent_buffer(C, monkeypatch):
    """The 1x1 conv hands mixdq_qlinear_f16in_w8a8 a fresh permute / reshape VIEW of its weight buffer on every
    call; the prefetch planner keeps weak references to what the entry points trace, and a reference to a
    temporary is dead on the next forward (the plan was dropped, silently: ADVICE r5).  What is traced is the
    module's buffer itself: alive, at the same address, on a second forward."""
    import weakref
    from mixdq_amd.nn import QuantizedConv2d
    from tests.cases import make_float_module
    from tests.test_host import prepared
    monkeypatch.setattr(C, "F16IN", "1")
    c = dict(key="ctr", kind="conv", name="up_blocks.0.resnets.2.conv_shortcut", cin=640, cout=1280, ksize=1,
             stride=1, pad=0, bias=True, seed=381, split=0)
    fm = make_float_module(c)
    w = fm.weight.detach().float()
    d = (w.reshape(1280, -1).abs().amax(dim=1) / 127).half()
    ck = {c["name"] + ".weight_quantizer": {"delta_list": d[None].repeat(3, 1), "zero_point_list": torch.zeros(3, 1280).half()},
          c["name"] + ".act_quantizer": {"delta_list": torch.tensor([0.05] * 3).half(),
                                         "zero_point_list": torch.tensor([120.0] * 3).half()}}
    qm = QuantizedConv2d.from_float(prepared(c, {}), split=0, ckpt=ck).to(DEV)
    x = torch.from_numpy(dd.f16(382, (1, 640, 32, 32), -3, 3)).to(DEV).contiguous(memory_format=torch.channels_last)
    refs = []
    for _ in range(2):
        ctx = C.PrefetchContext(DEV)
        with ctx, torch.no_grad():
            qm(x)
        assert len(ctx.trace) == 1 and ctx.trace[0] is qm.weight_int
        refs.append((weakref.ref(ctx.trace[0]), ctx.trace[0].data_ptr()))
        del ctx
    import gc
    gc.collect()
    assert all(r() is not None and r().data_ptr() == p for r, p in refs) and refs[0][1] == refs[1][1]
