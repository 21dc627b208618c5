"""GPU parity tests of the producer fusions (GroupNorm+SiLU+quantize, LayerNorm+quantize,
GEGLU+quantize): bit-exact vs the oracle's restatement (which fixes the reduction order and uses
the shared transcendental specification include/mixdq_math.h), and within one FP16 ulp of PyTorch's
fp32-reference GroupNorm / LayerNorm / SiLU / GELU (the ops the reference leaves to stock PyTorch;
one ulp per FP16 rounding point)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import detdata as dd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def scal(v):
    return torch.tensor(float(v), dtype=torch.float32, device=DEV)


def ulp_f16(ref):
    a = ref.float().abs().clamp(min=2.0 ** -14)
    return 2.0 ** (torch.floor(torch.log2(a)) - 10)


GN_CASES = [  # N, H, W, C, G, silu
    (1, 16, 16, 320, 32, True), (2, 8, 8, 640, 32, True), (1, 8, 8, 1280, 32, False),
    (1, 12, 12, 960, 32, True), (1, 4, 4, 1920, 32, True), (1, 4, 4, 2560, 32, True),
    (3, 5, 7, 64, 8, True), (2, 16, 16, 32, 8, False), (1, 1, 1, 128, 8, True),
    (1, 128, 128, 320, 32, True),
    # more than 64 partials per group: the sliced apply pass reduces its slice's groups itself (8 slices of 4
    # groups; 4 slices of 8 at C = 320 above) -- and a shape it cannot slice, which keeps the finalize launch
    (1, 32, 32, 640, 32, True), (2, 16, 16, 1280, 32, True), (1, 32, 32, 1920, 32, False),
    (1, 64, 64, 64, 8, True),
]


@pytest.mark.parametrize("case", GN_CASES, ids=[f"n{c[0]}_{c[1]}x{c[2]}_c{c[3]}_g{c[4]}_{'silu' if c[5] else 'plain'}" for c in GN_CASES])
def test_groupnorm_silu_quantize(C, oracle, case):
    N, H, W, Cc, G, silu = case
    x = (dd.normal_f16(11, (N, H, W, Cc), 1.5).astype(np.float32) +
         dd.normal_f16(12, (1, 1, 1, Cc), 0.7).astype(np.float32)).astype(np.float16)
    gamma = (dd.normal_f16(13, (Cc,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(14, (Cc,), 0.2)
    s_inv, zp = float(np.float32(1) / np.float32(0.031)), -20.0
    xd = t(x).permute(0, 3, 1, 2)     # NCHW view of NHWC memory
    q, h = C.groupnorm_silu_quantize(xd, G, t(gamma), t(beta), 1e-5, scal(s_inv), scal(zp),
                                     silu=silu, want_f16=True)
    assert q.shape == xd.shape and q.is_contiguous(memory_format=torch.channels_last) or N * H * W == 1 or Cc == 1
    q_ref, h_ref = oracle.groupnorm_silu_quantize(x, gamma, beta, 1e-5, G, silu, s_inv, zp,
                                                  C.FLAGS & 1)
    got_h = h.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    got_q = q.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    assert np.array_equal(got_h.view(np.uint16), h_ref.view(np.uint16)), \
        f"{(got_h.view(np.uint16) != h_ref.view(np.uint16)).sum()} fp16 values differ"
    assert np.array_equal(got_q, q_ref)
    # against PyTorch, one rounding point at a time (fp32 math, FP16 rounding after the norm and
    # after SiLU): the normalised value within 1 ulp of F.group_norm, and SiLU of OUR normalised
    # value within 1 ulp of F.silu of that same value (SiLU amplifies an input ulp for x << 0).
    _, pre = C.groupnorm_silu_quantize(xd, G, t(gamma), t(beta), 1e-5, silu=False, want_f16=True)
    ref = F.group_norm(xd.float(), G, t(gamma).float(), t(beta).float(), 1e-5).half()
    # (+2e-6: where a*x + b cancels to ~0 the FP32 rounding of the O(1) terms exceeds an FP16 ulp)
    assert ((pre.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref) + 2e-6).all()
    if silu:
        ref = F.silu(pre.float()).half()
        assert ((h.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref)).all()
    # int8-only and fp16-only variants agree with the combined call
    q2, none = C.groupnorm_silu_quantize(xd, G, t(gamma), t(beta), 1e-5, scal(s_inv), scal(zp),
                                         silu=silu)
    assert none is None and torch.equal(q2, q)
    none, h2 = C.groupnorm_silu_quantize(xd, G, t(gamma), t(beta), 1e-5, silu=silu, want_f16=True)
    assert none is None and torch.equal(h2, h)


def test_groupnorm_equals_unfused_pipeline_on_its_own_fp16(C):
    """quantize(fused fp16 output) == fused int8 output: the fusion only removes a round trip."""
    x = t(dd.normal_f16(21, (2, 8, 8, 640), 2.0)).permute(0, 3, 1, 2)
    g, b = t(dd.normal_f16(22, (640,), 1.0)), t(dd.normal_f16(23, (640,), 0.3))
    s_inv, zp = scal(17.3), scal(5.0)
    q, h = C.groupnorm_silu_quantize(x, 32, g, b, 1e-5, s_inv, zp, silu=True, want_f16=True)
    assert torch.equal(C.quantize_per_tensor_to_int8(h, s_inv, zp), q)


def test_groupnorm_sliced_apply_pass_same_bits():
    """MIXDQ_GN_SLICED=1 (the finalize launch folded into a channel-sliced apply pass: measured slower, off by
    default) is read once per process: the GroupNorm parity cases -- oracle, two sources, raw
    outputs -- again in a child process with the switch on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIXDQ_GN_SLICED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fused_gpu.py", "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k",
                        "test_groupnorm_silu_quantize or two_sources or raw_outputs"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("unroll", ["1", "4"])
def test_groupnorm_statistics_pixels_in_flight_same_bits(unroll):
    """The statistics pass keeps 4 pixels of a thread in flight (2 where a thread has no more than two;
    MIXDQ_GN_STATS_UNROLL=1: the plain loop, =4: four everywhere): the same additions in the same order -- the
    GroupNorm parity cases against the oracle again in a child process with the switch set (read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIXDQ_GN_STATS_UNROLL=unroll)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fused_gpu.py", "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k",
                        "test_groupnorm_silu_quantize or two_sources or raw_outputs"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]


def test_groupnorm_unsupported_shapes_raise(C):
    x = torch.zeros(1, 36, 4, 4, dtype=torch.float16, device=DEV).contiguous(
        memory_format=torch.channels_last)
    w = torch.ones(36, dtype=torch.float16, device=DEV)
    assert not C.groupnorm_supported(1, 16, 36, 6)
    with pytest.raises(RuntimeError, match="unsupported"):
        C.groupnorm_silu_quantize(x, 6, w, w, 1e-5, scal(1), scal(0))


LN_CASES = [(1024, 1280, 3), (4096, 640, 1), (77, 640, 2), (5, 64, 3), (3, 2048, 1), (1, 128, 0),
            (8193, 1280, 1), (8200, 640, 3), (8192, 128, 0)]     # two rows per wave (odd tail: one)


@pytest.mark.parametrize("M,Cc,nq", LN_CASES)
def test_layernorm_quantize(C, oracle, M, Cc, nq):
    x = (dd.normal_f16(31, (M, Cc), 1.2).astype(np.float32) + 0.4).astype(np.float16)
    gamma = (dd.normal_f16(32, (Cc,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(33, (Cc,), 0.2)
    qp = [(float(np.float32(1) / np.float32(0.02 + 0.01 * i)), float(-7 + 11 * i)) for i in range(nq)]
    outs, h = C.layernorm_quantize(t(x), t(gamma), t(beta), 1e-5,
                                   [(scal(a), scal(b)) for a, b in qp], want_f16=True)
    o_ref, h_ref = oracle.layernorm_quantize(x, gamma, beta, 1e-5, qp, C.FLAGS & 1)
    assert np.array_equal(h.cpu().numpy().view(np.uint16), h_ref.view(np.uint16))
    assert len(outs) == nq
    for a, b in zip(outs, o_ref):
        assert np.array_equal(a.cpu().numpy(), b)
    ref = F.layer_norm(t(x).float(), (Cc,), t(gamma).float(), t(beta).float(), 1e-5).half()
    assert ((h.float() - ref.float()).abs() <= 1.001 * ulp_f16(ref) + 2e-6).all()


@pytest.mark.parametrize("M,D", [(1024, 5120), (4096, 2560), (7, 64), (1, 8)])
def test_geglu_quantize(C, oracle, M, D):
    hin = dd.normal_f16(41, (M, 2 * D), 2.0)
    s_inv, zp = float(np.float32(1) / np.float32(0.05)), -100.0
    q, o = C.geglu_quantize(t(hin), scal(s_inv), scal(zp), want_f16=True)
    q_ref, o_ref = oracle.geglu_quantize(hin, s_inv, zp, C.FLAGS & 1)
    assert np.array_equal(o.cpu().numpy().view(np.uint16), o_ref.view(np.uint16))
    assert np.array_equal(q.cpu().numpy(), q_ref)
    hd = t(hin)
    ref = (hd[:, :D].float() * F.gelu(hd[:, D:].float()).half().float()).half()
    # a 1-ulp difference of the FP16 GELU value, times x, then rounded again: up to 2 ulp; plus,
    # for gate << 0, 1 + erf(gate / sqrt 2) cancels, so ANY FP32 erf (PyTorch's included) carries an
    # absolute error ~1e-7 * |gate| that is several ulps of the tiny result
    atol = 4e-7 * hd[:, :D].float().abs() * hd[:, D:].float().abs().clamp(min=1.0)
    assert ((o.float() - ref.float()).abs() <= 2.001 * ulp_f16(ref) + atol).all()


def test_shared_math_spec_matches_on_device(C, oracle):
    """include/mixdq_math.h evaluates identically under hipcc and gcc: drive SiLU / GELU through
    the fused kernels with identity normalisation and compare with the host evaluation."""
    L = oracle.lib()
    vals = np.concatenate([np.linspace(-12, 12, 4001), [0.0, -0.0, 65504, -65504, 1e-4, -1e-4]]
                          ).astype(np.float16)
    n = (vals.size + 7) // 8 * 8
    v = np.zeros(n, np.float16)
    v[:vals.size] = vals
    hin = np.concatenate([np.ones_like(v), v]).reshape(1, 2 * n)     # x = 1, gate = v
    _, o = C.geglu_quantize(t(hin), want_f16=True)
    with np.errstate(over="ignore"):
        want = np.array([np.float16(L.mixdq_oracle_geluf(float(a))) for a in v], np.float16)
    assert np.array_equal(o.cpu().numpy().reshape(-1).view(np.uint16), want.view(np.uint16))


def test_packed_gelu_equals_the_scalar_specification_on_every_fp16_gate(C, oracle):
    """The kernels evaluate GELU two values at a time on the packed-FP32 VALU (csrc/common.h
    geluf2); every half must perform the scalar specification's operations: all 65536 FP16 bit
    patterns as the gate (NaN / inf included), x = 1, against the host evaluation of
    include/mixdq_math.h -- and through the GEMM + GEGLU epilogue, whose INT8 output must be the
    quantizer applied to those values."""
    L = oracle.lib()
    v = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    hin = np.concatenate([np.ones_like(v), v]).reshape(1, 2 * v.size)
    _, o = C.geglu_quantize(t(hin), want_f16=True)
    with np.errstate(all="ignore"):
        want = np.array([np.float32(L.mixdq_oracle_geluf(float(a))) for a in v.astype(np.float32)],
                        np.float32).astype(np.float16)
    got = o.cpu().numpy().reshape(-1)
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(got.view(np.uint16)[~both_nan], want.view(np.uint16)[~both_nan])
    assert np.array_equal(np.isnan(got), np.isnan(want))


def test_gelu_table_of_the_large_tile_epilogue_is_the_specification(C, oracle):
    """The GEMM + GEGLU epilogue of the one-workgroup-per-CU tiles looks f16(gelu(g)) up in an LDS
    table for |g| < 16 and uses `g` (g >= 16) or `0 * g` (g <= -16: -0; -inf -> NaN) beyond: both
    halves against the host evaluation of include/mixdq_math.h, for all 65536 gates."""
    L = oracle.lib()
    tab = C.gelu_table(DEV).cpu().numpy().view(np.uint16)              # [2, 0x4c00]
    MAG = C.GELU_TABLE_MAG
    bits = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    g = bits.view(np.float16)
    with np.errstate(all="ignore"):
        want = np.array([np.float32(L.mixdq_oracle_geluf(float(a))) for a in g.astype(np.float32)],
                        np.float32).astype(np.float16)
        mag, neg = bits & 0x7fff, bits >> 15
        near = mag < MAG
        got = np.where(near, tab[neg, np.minimum(mag, MAG - 1)].view(np.float16),
                       np.where(neg == 1, (np.float32(0) * g.astype(np.float32)).astype(np.float16), g))
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got.view(np.uint16)[~both_nan], want.view(np.uint16)[~both_nan])


GEGLU_GEMM_CASES = [  # M, D, K, forced tile config (0 = automatic), bias
    (1024, 640, 320, 0, True), (96, 64, 64, 4, True), (77, 32, 48, 4, False),
    (200, 160, 128, 3, True), (200, 160, 128, 41, True), (300, 128, 256, 13, True),
    (513, 256, 192, 18, False), (257, 256, 128, 20, True), (1, 32, 16, 0, True),
    (300, 320, 256, 25, True), (300, 320, 192, 46, True), (130, 160, 128, 46, False),
    # the four-phase 256x256 loop: M tail, several column tiles, a K tail (one-phase fallback)
    (300, 256, 256, 70, True), (513, 384, 384, 70, False), (257, 256, 192, 70, True),
    # ... and its persistent form (one tile per workgroup at these sizes; the tile loop: test_persistent_... below)
    (300, 256, 256, 71, True), (513, 384, 384, 71, False), (257, 256, 192, 71, True),
    # every remaining tile family: 64x64x64, the k-split 64x64, 128x128 8-wave, 128x256, 256x256,
    # the deeper 128x320 pipelines; D = 16 (one group)
    (96, 64, 64, 1, True), (130, 64, 256, 37, True), (200, 128, 128, 35, False), (300, 128, 128, 15, True),
    (257, 256, 128, 14, True), (300, 320, 320, 47, True), (40, 16, 64, 0, True),
]


@pytest.mark.parametrize("case", GEGLU_GEMM_CASES, ids=[f"m{c[0]}_d{c[1]}_k{c[2]}_cfg{c[3]}" for c in GEGLU_GEMM_CASES])
def test_qlinear_geglu_equals_gemm_then_geglu_quantize(C, oracle, case):
    """GEMM + GEGLU + quantize in one launch (value/gate-interleaved weight rows) is bit-identical
    to the oracle's qlinear followed by its geglu_quantize on the ordinary row order."""
    M, D, K, cfg, has_bias = case
    a = dd.int8(51, (M, K))
    w = dd.int8(52, (2 * D, K))
    scale = dd.f32(53, (2 * D,), 2e-4, 9e-4)
    bias0 = (dd.f32(54, (2 * D,), -300, 300)).astype(np.float32)
    bias = dd.normal_f16(55, (2 * D,), 0.5) if has_bias else None
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    h = oracle.qlinear(a, w, bias0, scale, bias, C.FLAGS & 1)
    q_ref, _ = oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)
    perm = C.geglu_row_order(D, DEV)
    wd, sd, bd = t(w)[perm].contiguous(), t(scale)[perm].contiguous(), t(bias0)[perm].contiguous()
    biasd = t(bias)[perm].contiguous() if has_bias else None
    got = C.qlinear_geglu(t(a), wd, sd, bd, biasd, scal(s_inv), scal(zp), _cfg=cfg)
    assert got.shape == (M, D) and got.dtype == torch.int8
    assert np.array_equal(got.cpu().numpy(), q_ref)
    assert q_ref.size < 4096 or len(np.unique(q_ref)) > 16   # the case exercises the int8 range
    # and equals the two-launch HIP chain on the ordinary row order
    hd = C.qlinear_w8_a8_ohalf(t(a), t(w), t(scale), scal(1), scal(0), t(bias0), t(scale), t(bias0),
                               None if bias is None else t(bias))
    q2, _ = C.geglu_quantize(hd, scal(s_inv), scal(zp))
    assert torch.equal(got, q2)


@pytest.mark.parametrize("cfg", [70, 71, 25, 13])
def test_qlinear_geglu_into_an_8_byte_aligned_output(C, oracle, cfg):
    """The entry point asks for 8-byte alignment of the INT8 output only; the register epilogue of the
    256x256 tile stores 16 bytes at a time when it can and 2 x 8 otherwise.  Same bits either way."""
    M, D, K = 300, 320, 256
    a, w = dd.int8(61, (M, K)), dd.int8(62, (2 * D, K))
    scale, bias0 = dd.f32(63, (2 * D,), 2e-4, 9e-4), dd.f32(64, (2 * D,), -300, 300)
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    perm = C.geglu_row_order(D, DEV)
    args = (t(a), t(w)[perm].contiguous(), t(scale)[perm].contiguous(), t(bias0)[perm].contiguous(), None,
            scal(s_inv), scal(zp))
    want = C.qlinear_geglu(*args, _cfg=cfg)
    buf = torch.full((M * D + 24,), 77, dtype=torch.int8, device=DEV)
    out = buf[8:8 + M * D].view(M, D)
    assert out.data_ptr() % 16 == 8
    got = C.qlinear_geglu(*args, _cfg=cfg, _out=out)
    assert got.data_ptr() == out.data_ptr() and torch.equal(got, want)
    assert (buf[:8] == 77).all() and (buf[8 + M * D:] == 77).all()      # nothing written outside
    h = oracle.qlinear(a, w, bias0, scale, None, C.FLAGS & 1)
    assert np.array_equal(want.cpu().numpy(), oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)[0])


@pytest.mark.parametrize("cfg", [70, 71, 25, 13, 0])
def test_qlinear_geglu_with_overflowing_and_nan_columns(C, oracle, cfg):
    """Gates and values beyond the GELU table's range, +-inf (fp16 overflow of the GEMM output) and NaN
    (a NaN bias0): the table epilogues select g / -0 / NaN on the bits there -- same INT8 tensor as
    the oracle's chain, in the same launch as ordinary columns."""
    M, D, K = 300, 320, 256
    a, w = dd.int8(71, (M, K)), dd.int8(72, (2 * D, K))
    scale, bias0 = dd.f32(73, (2 * D,), 2e-4, 9e-4), dd.f32(74, (2 * D,), -300, 300)
    scale[D + 3], scale[D + 40], scale[7], scale[D + 100] = 10.0, 1e-2, 10.0, 4e-3   # gate inf / large, value inf
    bias0[D + 11], bias0[19] = np.nan, np.nan                                         # NaN gate, NaN value
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    with np.errstate(all="ignore"):
        h = oracle.qlinear(a, w, bias0, scale, None, C.FLAGS & 1)
        q_ref, _ = oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)
    assert np.isinf(h[:, D + 3]).any() and np.isnan(h[:, D + 11]).all() and (np.abs(h[:, D + 40]) > 16).any()
    perm = C.geglu_row_order(D, DEV)
    got = C.qlinear_geglu(t(a), t(w)[perm].contiguous(), t(scale)[perm].contiguous(),
                          t(bias0)[perm].contiguous(), None, scal(s_inv), scal(zp), _cfg=cfg)
    assert np.array_equal(got.cpu().numpy(), q_ref)


def test_qlinear_geglu_rejects_tiles_without_whole_value_gate_groups(C):
    """BN % 32 != 0 tiles (the 16x16x64-MFMA exact-fit tiles) cannot hold whole value|gate groups
    (16 + 16 columns): forcing one is an error, not a wrong result."""
    a = torch.zeros(64, 128, dtype=torch.int8, device=DEV)
    w = torch.zeros(320, 128, dtype=torch.int8, device=DEV)
    v = torch.ones(320, device=DEV)
    with pytest.raises(RuntimeError, match="N % 32"):
        C.qlinear_geglu(a, w, v, v, None, scal(1.0), scal(0.0), _cfg=42)
    with pytest.raises(RuntimeError, match="N % 32"):        # 128x320 on 4 x 4 waves: 80 columns per wave
        C.qlinear_geglu(a, w, v, v, None, scal(1.0), scal(0.0), _cfg=28)


def test_qlinear_geglu_argument_checks(C):
    a = torch.zeros(8, 64, dtype=torch.int8, device=DEV)
    w = torch.zeros(80, 64, dtype=torch.int8, device=DEV)      # N = 80: not whole groups of 32
    v = torch.zeros(80, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError):
        C.qlinear_geglu(a, w, v, v, None, scal(1), scal(0))


@pytest.mark.parametrize("M,D,K,cfg", [(200, 160, 128, 0), (96, 64, 64, 4), (300, 128, 256, 3)])
def test_qlinear_geglu_w4_equals_oracle_chain_on_unpacked_weights(C, oracle, M, D, K, cfg):
    """Packed-W4 weights through the fused GEMM + GEGLU + quantize launch: the oracle's W8 qlinear on
    oracle.unpack_w4(packed), then its geglu_quantize, bit for bit."""
    from mixdq_amd.nn.utils import pack_w4
    a = dd.int8(61, (M, K))
    q4 = dd.int8(62, (2 * D, K), -8, 8)
    scale = dd.f32(63, (2 * D,), 2e-3, 9e-3)
    bias0 = dd.f32(64, (2 * D,), -30, 30).astype(np.float32)
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    packed = pack_w4(torch.from_numpy(q4))
    assert np.array_equal(oracle.unpack_w4(packed.numpy()), q4)
    h = oracle.qlinear(a, q4, bias0, scale, None, C.FLAGS & 1)
    q_ref, _ = oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)
    perm = C.geglu_row_order(D, DEV)
    got = C.qlinear_geglu(t(a), packed.to(DEV)[perm].contiguous(), t(scale)[perm].contiguous(),
                          t(bias0)[perm].contiguous(), None, scal(s_inv), scal(zp), _cfg=cfg, _w4=True)
    assert np.array_equal(got.cpu().numpy(), q_ref)


# ----------------------------------------------- to_q + cross-attention in one launch (f-1)
QATT_CASES = [  # B, T (query rows per image), C = heads * 64, K, Tkv, packed k|v?, quantized out?
    (1, 1024, 1280, 1280, 77, True, True), (2, 256, 640, 640, 77, True, True),
    (1, 64, 128, 128, 77, False, False), (1, 128, 256, 384, 128, False, True),
    (3, 64, 128, 256, 64, True, False), (1, 192, 256, 128, 13, False, True),
    (1, 4096, 640, 640, 77, True, True),
]


@pytest.mark.parametrize("case", QATT_CASES, ids=[f"b{c[0]}_t{c[1]}_c{c[2]}_k{c[3]}_kv{c[4]}" for c in QATT_CASES])
def test_qlinear_attention_equals_to_q_then_attention(C, oracle, case):
    """mixdq_qlinear_w8a8_attn == mixdq_qlinear_w8a8 (to_q, bit-exact vs the oracle elsewhere)
    followed by mixdq_attention_f16, bit for bit -- fp16 output and the fused INT8 output -- and
    within the attention tolerance of the float64 oracle."""
    B, T, Cc, K, Tkv, packed, quant = case
    a = dd.int8(201, (B, T, K))
    w = dd.int8(202, (Cc, K))
    scale, bias0 = dd.f32(203, (Cc,), 1e-5, 6e-5), dd.f32(204, (Cc,), -300, 300)
    if packed:   # column slices of a packed k|v projection, BOS-style rows
        kv = t(dd.normal_f16(205, (B, Tkv, 2 * Cc), 1.0))
        k, v = kv[..., :Cc], kv[..., Cc:]
    else:
        k, v = t(dd.normal_f16(206, (B, Tkv, Cc), 1.0)), t(dd.normal_f16(207, (B, Tkv, Cc), 1.0))
    s_inv, zp = scal(30.0), scal(-3.0)
    q = C.qlinear_w8_a8_ohalf(t(a), t(w), t(scale), scal(1), scal(0), t(bias0), t(scale), t(bias0), None)
    want_h = C.attention_f16(q, k, v, Cc // 64)
    got_h = C.qlinear_attention(t(a), t(w), t(scale), t(bias0), k, v)
    assert got_h.dtype == torch.float16 and tuple(got_h.shape) == (B, T, Cc)
    assert torch.equal(got_h, want_h), f"{(got_h != want_h).sum().item()} of {got_h.numel()} differ"
    if quant:
        want_q = C.attention_f16(q, k, v, Cc // 64, s_inv, zp)
        got_q = C.qlinear_attention(t(a), t(w), t(scale), t(bias0), k, v, s_inv, zp)
        assert got_q.dtype == torch.int8 and torch.equal(got_q, want_q)
    if B * T <= 1024:
        _, ref = oracle.attention_f16(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy(), Cc // 64)
        err = np.abs(got_h.cpu().numpy().astype(np.float64) - ref)
        assert (err <= 2e-3 + 4e-3 * np.abs(ref)).all(), err.max()


def test_qlinear_attention_w4_and_argument_checks(C):
    from mixdq_amd.nn.utils import pack_w4
    B, T, Cc, K, Tkv = 1, 128, 128, 256, 77
    a = t(dd.int8(211, (B, T, K)))
    qw = dd.int8(212, (Cc, K), -8, 8)
    scale, bias0 = t(dd.f32(213, (Cc,), 1e-4, 6e-4)), t(dd.f32(214, (Cc,), -30, 30))
    k, v = t(dd.normal_f16(215, (B, Tkv, Cc), 1.0)), t(dd.normal_f16(216, (B, Tkv, Cc), 1.0))
    q = C.qlinear_w8_a8_ohalf(a, t(qw), scale, scal(1), scal(0), bias0, scale, bias0, None)
    want = C.attention_f16(q, k, v, 2, scal(20.0), scal(1.0))
    got = C.qlinear_attention(a, pack_w4(torch.from_numpy(qw)).to(DEV), scale, bias0, k, v, scal(20.0),
                              scal(1.0), _w4=True)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError, match="unsupported configuration"):
        C.qlinear_attention(a[:, :100], t(qw), scale, bias0, k, v)            # T % 64
    k200 = t(dd.normal_f16(217, (B, 200, Cc), 1.0))
    with pytest.raises(RuntimeError, match="unsupported configuration"):
        C.qlinear_attention(a, t(qw), scale, bias0, k200, k200)               # Tkv > 128


# ------------------------------------ up-block skip connections without the concatenation (f-1)
@pytest.mark.parametrize("N,H,W,C1,C2,G,silu", [(1, 16, 16, 64, 32, 8, True), (2, 8, 12, 32, 64, 8, False),
                                                (1, 32, 32, 640, 320, 32, True), (1, 9, 7, 24, 8, 4, True)])
def test_groupnorm_two_sources_equals_concatenation(C, N, H, W, C1, C2, G, silu):
    """GroupNorm over cat([x, x2], dim=1) read from the two tensors in place == the same kernel on
    the concatenated tensor (itself bit-exact vs the oracle above), int8 and fp16 outputs."""
    xa = t(dd.normal_f16(301, (N, H, W, C1), 1.3)).permute(0, 3, 1, 2)      # channels-last
    xb = t(dd.normal_f16(302, (N, H, W, C2), 0.7)).permute(0, 3, 1, 2)
    gamma = t((dd.normal_f16(303, (C1 + C2,), 0.3).astype(np.float32) + 1).astype(np.float16))
    beta = t(dd.normal_f16(304, (C1 + C2,), 0.2))
    s_inv, zp = scal(25.0), scal(-9.0)
    cat = torch.cat([xa, xb], dim=1).contiguous(memory_format=torch.channels_last)
    q1, h1 = C.groupnorm_silu_quantize(cat, G, gamma, beta, 1e-5, s_inv, zp, silu=silu, want_f16=True)
    q2, h2 = C.groupnorm_silu_quantize(xa, G, gamma, beta, 1e-5, s_inv, zp, silu=silu, want_f16=True,
                                       x2=xb)
    assert q2.shape == q1.shape and q2.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(q1, q2) and torch.equal(h1, h2)


@pytest.mark.parametrize("N,H,W,C1,C2,G,which", [(1, 16, 16, 64, 32, 8, (1, 1)), (2, 8, 12, 32, 64, 8, (0, 1)),
                                                 (1, 32, 32, 640, 320, 32, (1, 1)), (1, 9, 7, 24, 0, 4, (1,)),
                                                 (2, 32, 32, 320, 0, 32, (1,)), (1, 9, 7, 24, 8, 4, (1, 0))])
def test_groupnorm_raw_outputs_are_the_quantized_inputs(C, oracle, N, H, W, C1, C2, G, which):
    """The apply pass's second product: each source tensor quantized as it is (the ResNet
    shortcut's operand) == mixdq_quantize_f16_i8 of that tensor == the oracle, with the norm's own
    outputs unchanged."""
    xa = t(dd.normal_f16(311, (N, H, W, C1), 1.3)).permute(0, 3, 1, 2)
    xb = t(dd.normal_f16(312, (N, H, W, C2), 0.7)).permute(0, 3, 1, 2) if C2 else None
    gamma = t((dd.normal_f16(313, (C1 + C2,), 0.3).astype(np.float32) + 1).astype(np.float16))
    beta = t(dd.normal_f16(314, (C1 + C2,), 0.2))
    s_inv, zp = scal(25.0), scal(-9.0)
    qps = [(scal(40.0), scal(11.0)), (scal(17.5), scal(-30.0))][:len(which)]
    raw_qp = [qp if w else None for qp, w in zip(qps, which)]
    q0, h0 = C.groupnorm_silu_quantize(xa, G, gamma, beta, 1e-5, s_inv, zp, want_f16=True, x2=xb)
    q1, h1, raws = C.groupnorm_silu_quantize(xa, G, gamma, beta, 1e-5, s_inv, zp, want_f16=True, x2=xb,
                                             raw_qparams=raw_qp)
    assert torch.equal(q0, q1) and torch.equal(h0, h1)
    for src, qp, w, r in zip([xa, xb], qps, which, raws):
        if not w:
            assert r is None
            continue
        assert r.shape == src.shape and r.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(r, C.quantize_per_tensor_to_int8(src, qp[0], qp[1]))
        want = oracle.quantize(src.permute(0, 2, 3, 1).contiguous().cpu().numpy(),
                               float(qp[0]), float(qp[1]), C.FLAGS & 1)
        assert np.array_equal(r.permute(0, 2, 3, 1).contiguous().cpu().numpy(), want)
    # only the f16 output, no consumer quantizer
    _, h2, raws2 = C.groupnorm_silu_quantize(xa, G, gamma, beta, 1e-5, want_f16=True, x2=xb,
                                             raw_qparams=raw_qp)
    assert torch.equal(h2, h0) and all((a is None) == (b is None) and (a is None or torch.equal(a, b))
                                       for a, b in zip(raws, raws2))
    with pytest.raises(RuntimeError):
        C.groupnorm_silu_quantize(xa, G, gamma, beta, 1e-5, s_inv, zp, x2=xb, raw_qparams=[qps[0]] * 3)


def test_split_shortcut_on_unconcatenated_halves(C, modules_golden):
    """QuantizedConv2d.forward_parts(x_a, x_b) == forward(cat([x_a, x_b])) == the reference class's
    output (modules.npz), bit for bit."""
    from mixdq_amd.nn import QuantizedConv2d
    from tests.cases import MODULE_CASES, module_ckpt, module_input
    from tests.test_host import prepared
    c = next(c for c in MODULE_CASES if c["key"] == "conv_split")
    qm = QuantizedConv2d.from_float(prepared(c, modules_golden), split=c["split"],
                                    ckpt=module_ckpt(c, modules_golden)).to(DEV)
    x = module_input(c).to(DEV).contiguous(memory_format=torch.channels_last)
    xa = x[:, :c["split"]].contiguous(memory_format=torch.channels_last)
    xb = x[:, c["split"]:].contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y, yp = qm(x), qm.forward_parts(xa, xb)
        ypq = qm.forward_parts_quantized(C.quantize_per_tensor_to_int8(xa, qm.act_scales_inv, qm.act_zero_points),
                                         C.quantize_per_tensor_to_int8(xb, qm.act_scales_inv_0,
                                                               qm.act_zero_points_0))
    assert torch.equal(y, yp) and torch.equal(y, ypq)
    want = modules_golden["conv_split.out"]
    assert np.array_equal(yp.contiguous().cpu().numpy().view(np.uint16), want.view(np.uint16))


# --------------------------------------------------------- GEMM + residual + LayerNorm + quantize (round 5)
LN_GEMM_CASES = [  # M, N, K, residual, bias, nq, want_f16, expected tile id
    (1024, 1280, 1280, True, True, 1, False, 56),      # attn.to_out.0 -> norm2 / norm3
    (1024, 1280, 1280, True, True, 3, True, 56),       # ff.net.2 -> the next block's norm1 (q, k, v quantizers)
    (1024, 1280, 5120, True, True, 1, False, 45),      # ff.net.2 at 1280 channels
    (1024, 1280, 1280, False, True, 2, False, 56),     # proj_in -> norm1 (no residual)
    (4096, 640, 640, True, True, 1, False, 44),
    (4096, 640, 2560, True, False, 3, False, 44),
    (2048, 1280, 1280, True, True, 1, True, 44),       # batch 2 at 1280 channels: 128-row tiles
    (1000, 1280, 1280, True, True, 2, False, 56),      # ragged last row tile
    (77, 640, 640, True, True, 0, True, 56),           # FP16 copy only
    (1024, 1280, 1280, True, False, 2, True, 56),
]


@pytest.mark.parametrize("M,N,K,res,bias,nq,want_h,cfg", LN_GEMM_CASES)
def test_qlinear_ln_equals_gemm_then_layernorm_quantize(C, oracle, M, N, K, res, bias, nq, want_h, cfg):
    """mixdq_qlinear_w8a8_ln (csrc/igemm_ln.hip): ONE launch == mixdq_qlinear_w8a8_rows (+ residual) followed by
    mixdq_layernorm_quantize, bit for bit -- the FP16 rows, every INT8 tensor, the FP16 LayerNorm copy -- and
    == the oracle's chain.  The launch runs twice on one exchange buffer (records are told apart by the launch
    tag the buffer counts up), and once more after a launch of another shape."""
    assert C._lib.mixdq_qlinear_ln_select_id(M, N, K) == (cfg if K <= 1280 else -1)   # long K: two launches by rule
    a, w = dd.int8(701, (M, K)), dd.int8(702, (N, K))
    b0, sc = dd.f32(703, (N,), -500, 500), dd.f32(704, (N,), 2e-5, 6e-5)
    bs = dd.f16(705, (N,), -1, 1) if bias else None
    r = dd.normal_f16(706, (M, N), 1.5) if res else None
    gamma = (dd.normal_f16(707, (N,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(708, (N,), 0.2)
    qp = [(float(np.float32(1) / np.float32(0.02 + 0.01 * i)), float(-7 + 11 * i)) for i in range(nq)]
    qpd = [(scal(x), scal(y)) for x, y in qp]
    ws = C.qlinear_ln_workspace(max(M, 2048), N, DEV)
    args = (t(a), t(w), t(sc), t(b0), None if bs is None else t(bs), None if r is None else t(r), t(gamma),
            t(beta), 1e-5, qpd, ws)
    y, outs, h = C.qlinear_ln(*args, want_f16=want_h, _cfg=cfg)
    # the two launches it stands for
    y2 = C.qlinear_w8_a8_ohalf(t(a), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0),
                               None if bs is None else t(bs), _residual=None if r is None else t(r))
    o2, h2 = C.layernorm_quantize(y2, t(gamma), t(beta), 1e-5, qpd, want_f16=want_h)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    assert len(outs) == nq and all(torch.equal(p, q) for p, q in zip(outs, o2))
    if want_h:
        assert torch.equal(h.view(torch.int16), h2.view(torch.int16))
    # the oracle's chain
    v = C.FLAGS & 1
    y_ref = oracle.qlinear(a, w, b0, sc, bs, v)
    if r is not None:
        y_ref = oracle.add_f16(y_ref, r)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), y_ref.view(np.uint16))
    o_ref, h_ref = oracle.layernorm_quantize(y_ref, gamma, beta, 1e-5, qp, v)
    for p, q in zip(outs, o_ref):
        assert np.array_equal(p.cpu().numpy(), q)
    if want_h:
        assert np.array_equal(h.cpu().numpy().view(np.uint16), h_ref.view(np.uint16))
    # the exchange buffer is clean again: same launch, another shape, same launch
    hdr = ws[:8].view(torch.int32).tolist()                        # (epoch, departures): one launch, all gone
    assert hdr == [1, 0], hdr
    y3, outs3, _ = C.qlinear_ln(*args, want_f16=want_h, _cfg=cfg)
    other = C.qlinear_ln(t(dd.int8(711, (128, 640))), t(dd.int8(712, (640, 640))), t(sc[:640].copy()),
                         t(b0[:640].copy()), None, None, t(gamma[:640].copy()), t(beta[:640].copy()), 1e-5,
                         qpd[:1] or [(scal(30), scal(0))], ws)
    assert other[0].shape == (128, 640)
    y4, outs4, _ = C.qlinear_ln(*args, want_f16=want_h, _cfg=cfg)
    for yy, oo in ((y3, outs3), (y4, outs4)):
        assert torch.equal(yy.view(torch.int16), y2.view(torch.int16))
        assert all(torch.equal(p, q) for p, q in zip(oo, o2))


def test_qlinear_ln_range_and_graph_capture(C):
    """Outside its range the launch says so (the caller then issues the two launches); inside a captured graph
    it replays (no host synchronisation, the counters reset themselves)."""
    assert not C.qlinear_ln_supported(8192, 1280, 1280)        # more tiles than CUs: not all resident at once
    assert not C.qlinear_ln_supported(1024, 1280, 200)
    assert not C.qlinear_ln_supported(1024, 1920, 1280)        # a column tile must be one unit of the LayerNorm
    assert C.qlinear_ln_supported(1024, 320, 1280)             # (N = 320, 640, 1280: 4, 8, 16 units of 80)
    assert C.qlinear_ln_supported(1024, 1280, 1280) and C.qlinear_ln_supported(4096, 640, 640)
    assert not C.qlinear_ln_supported(4096, 640, 2560)         # long K: the cost rule keeps two launches
    M, N, K = 1024, 1280, 1280
    a, w = t(dd.int8(721, (M, K))), t(dd.int8(722, (N, K)))
    sc, b0 = t(dd.f32(723, (N,), 2e-5, 6e-5)), t(dd.f32(724, (N,), -50, 50))
    r = t(dd.normal_f16(725, (M, N), 1.0))
    g, b = t(dd.normal_f16(726, (N,), 1.0)), t(dd.normal_f16(727, (N,), 0.2))
    qp = [(scal(40), scal(3))]
    ws = C.qlinear_ln_workspace(M, N, DEV)
    with pytest.raises(RuntimeError, match="shape outside"):
        C.qlinear_ln(t(dd.int8(728, (8192, K))), w, sc, b0, None, None, g, b, 1e-5, qp,
                     C.qlinear_ln_workspace(8192, N, DEV))
    ref = C.qlinear_ln(a, w, sc, b0, None, r, g, b, 1e-5, qp, ws)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = C.qlinear_ln(a, w, sc, b0, None, r, g, b, 1e-5, qp, ws)
    for _ in range(3):
        got[0].zero_()
        got[1][0].zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got[0].view(torch.int16), ref[0].view(torch.int16))
        assert torch.equal(got[1][0], ref[1][0])
    # every launch above found all its records: the workspace's sticky error word is clean (a workgroup that
    # gives up waiting writes NaN rows AND leaves its launch tag there -- ADVICE r5); the epoch has grown
    assert C.qlinear_ln_status(ws) == 0
    assert int(ws[:4].view(torch.int32).item()) >= 4
    ws[8:12].view(torch.int32).fill_(77)                        # (the word is the host's to read, not a launch's to clear)
    C.qlinear_ln(a, w, sc, b0, None, r, g, b, 1e-5, qp, ws)
    assert C.qlinear_ln_status(ws) == 77


PP_CASES = [  # M, N, K, workgroups the grid is capped at, GEGLU?, residual?, bias?
    (1100, 768, 640, 8, True, False, True),      # 15 tiles on 8 workgroups, 5 K-tiles: the buffer parity flips per tile
    (1100, 768, 512, 8, True, False, False),     # 4 K-tiles: the parity stays
    (2100, 1024, 640, 16, True, False, True),    # 36 tiles on 16 workgroups: runs of 5 / 4 tiles per XCD, 2 - 3 per workgroup
    (1100, 768, 640, 8, False, False, True),
    (1100, 776, 512, 8, False, True, True),      # plain epilogue with a residual, N % 256 != 0 and N % 16 != 0
    (2100, 1024, 384, 16, False, True, False),
    (700, 512, 256, 8, False, False, False),     # two K-tiles: the shortest main loop the form takes
]


@pytest.mark.parametrize("M,N,K,wgs,geglu,res,bias", PP_CASES)
def test_persistent_tile_loop_same_bits_with_several_tiles_per_workgroup(C, oracle, monkeypatch, M, N, K, wgs, geglu,
                                                                         res, bias):
    """The persistent four-phase kernel (configuration 71, csrc/igemm_pp.h) with its grid capped so that every
    workgroup walks two or three tiles: the next tile's first K-tile staged during the current tile's last one and
    landing under its epilogue, the epilogue's LDS region alternating between the two stage buffers when the
    K-tile count is odd, the per-tile parameter block -- bit-identical to the oracle and to the
    one-tile-per-workgroup kernel (70)."""
    monkeypatch.setenv("MIXDQ_IGEMM_PERSIST_WGS", str(wgs))
    a, w = dd.int8(811, (M, K)), dd.int8(812, (N, K))
    scale, bias0 = dd.f32(813, (N,), 2e-4, 9e-4), dd.f32(814, (N,), -300, 300)
    bs = dd.normal_f16(815, (N,), 0.5) if bias else None
    if geglu:
        D = N // 2
        s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
        perm = C.geglu_row_order(D, DEV)
        args = (t(a), t(w)[perm].contiguous(), t(scale)[perm].contiguous(), t(bias0)[perm].contiguous(),
                None if bs is None else t(bs)[perm].contiguous(), scal(s_inv), scal(zp))
        got = C.qlinear_geglu(*args, _cfg=71)
        h = oracle.qlinear(a, w, bias0, scale, bs, C.FLAGS & 1)
        assert np.array_equal(got.cpu().numpy(), oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)[0])
        assert torch.equal(got, C.qlinear_geglu(*args, _cfg=70))
        return
    r = dd.normal_f16(816, (M, N), 1.5) if res else None
    kw = dict(_residual=t(r)) if res else {}
    got = C.qlinear_w8_a8_ohalf(t(a), t(w), t(scale), scal(1), scal(0), t(bias0), t(scale), t(bias0),
                                None if bs is None else t(bs), _cfg=71, **kw)
    want = oracle.qlinear(a, w, bias0, scale, bs, C.FLAGS & 1)
    if res:
        want = oracle.add_f16(want, r)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want.view(np.uint16))
    ref70 = C.qlinear_w8_a8_ohalf(t(a), t(w), t(scale), scal(1), scal(0), t(bias0), t(scale), t(bias0),
                                  None if bs is None else t(bs), _cfg=70, **kw)
    assert torch.equal(got.view(torch.int16), ref70.view(torch.int16))


def test_silu_table_is_the_specification(C, oracle):
    """The GroupNorm apply pass of large launches looks f16(silu(y)) up in an LDS table (csrc/fused_norm.hip, round 6):
    every entry against the host evaluation of include/mixdq_math.h; and the ranges it leaves to the arithmetic are the
    trivial ones of the specification (silu(y) == y above, -0 below) -- not relied on by the kernel (values outside the
    table take the arithmetic itself), but it is why the table ends where it ends."""
    import ctypes
    L = oracle.lib()
    L.mixdq_oracle_siluf.restype, L.mixdq_oracle_siluf.argtypes = ctypes.c_float, [ctypes.c_float]
    tab, n_pos, n_neg = C.silu_table(DEV)
    tab = tab.cpu().numpy().view(np.uint16)
    assert tab.size == n_pos + n_neg and n_pos == 0x4810 and n_neg == 0x4d20
    bits = np.concatenate([np.arange(n_pos, dtype=np.uint16), (0x8000 | np.arange(n_neg, dtype=np.uint32)).astype(np.uint16)])
    with np.errstate(all="ignore"):
        want = np.array([np.float32(L.mixdq_oracle_siluf(float(a))) for a in bits.view(np.float16).astype(np.float32)],
                        np.float32).astype(np.float16)
    assert np.array_equal(tab, want.view(np.uint16))
    rest_pos = np.arange(n_pos, 0x7c01, dtype=np.uint16)                       # up to +inf
    rest_neg = (0x8000 | np.arange(n_neg, 0x7c00, dtype=np.uint32)).astype(np.uint16)
    with np.errstate(all="ignore"):
        sp = np.array([np.float32(L.mixdq_oracle_siluf(float(a))) for a in rest_pos.view(np.float16).astype(np.float32)],
                      np.float32).astype(np.float16)
        sn = np.array([np.float32(L.mixdq_oracle_siluf(float(a))) for a in rest_neg.view(np.float16).astype(np.float32)],
                      np.float32).astype(np.float16)
    assert np.array_equal(sp.view(np.uint16), rest_pos) and (sn.view(np.uint16) == 0x8000).all()


def test_groupnorm_silu_table_variant_same_bits():
    """MIXDQ_GN_SILU_TAB=1 (read once per process) sends every GroupNorm + SiLU launch with a finalize launch through
    the table variant of the apply pass -- persistent blocks, SiLU by LDS lookup, values beyond the table by the
    arithmetic -- instead of only the large ones: the GroupNorm parity cases (oracle, two sources, raw outputs, and the
    case below with values far beyond the table) again in a child process with the switch on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIXDQ_GN_SILU_TAB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fused_gpu.py", "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k",
                        "test_groupnorm_silu_quantize or two_sources or raw_outputs or beyond_the_silu_table"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("N,H,W,Cc", [(1, 32, 32, 640), (2, 64, 64, 320)])
def test_groupnorm_silu_values_beyond_the_silu_table(C, oracle, N, H, W, Cc):
    """gamma = +-12: normalised values reach +-40 -- above 8.06 SiLU rounds to the value itself, below -20.5 to -0, and
    the table variant of the apply pass (the child-process test above; here whichever variant the launch takes) computes
    those by the arithmetic: bit-exact vs the oracle, INT8 and FP16 outputs."""
    x = dd.normal_f16(911, (N, H, W, Cc), 1.5)
    gamma = (np.where(np.arange(Cc) % 2 == 0, 12.0, -12.0)).astype(np.float16)
    beta = dd.normal_f16(912, (Cc,), 0.5)
    s_inv, zp = float(np.float32(1) / np.float32(0.3)), -20.0
    xd = t(x).permute(0, 3, 1, 2)
    q, h = C.groupnorm_silu_quantize(xd, 32, t(gamma), t(beta), 1e-5, scal(s_inv), scal(zp), silu=True, want_f16=True)
    q_ref, h_ref = oracle.groupnorm_silu_quantize(x, gamma, beta, 1e-5, 32, True, s_inv, zp, C.FLAGS & 1)
    got_h = h.permute(0, 2, 3, 1).contiguous().cpu().numpy()
    assert np.abs(h_ref.astype(np.float32)).max() > 25 and (h_ref.view(np.uint16) == 0x8000).any()
    assert np.array_equal(got_h.view(np.uint16), h_ref.view(np.uint16))
    assert np.array_equal(q.permute(0, 2, 3, 1).contiguous().cpu().numpy(), q_ref)


def test_geglu_table_variant_same_bits():
    """MIXDQ_GEGLU_TAB=1 (read once per process) sends every stand-alone GEGLU + quantize launch through the table
    variant (csrc/fused_norm.hip geglu_quant_tab_kernel: GELU looked up in LDS, gates beyond +-16 / NaN / inf by the
    arithmetic) instead of only the launches of 2 Mi outputs and more: the GEGLU parity cases -- the oracle, the
    specification on 4 000 gates, ALL 65 536 FP16 gates -- again in a child process with the switch on, and once more
    with it off (the large cases then take the arithmetic kernel the small ones always take)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fused_gpu.py", "-m", "gpu", "-q", "-x",
                            "-p", "no:cacheprovider", "-k",
                            "test_geglu_quantize or shared_math_spec or every_fp16_gate"],
                           cwd=root, env=dict(os.environ, MIXDQ_GEGLU_TAB=flag), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:]
