"""GPU tests of the FP16 attention core (mixdq_attention_f16).

The reference leaves the attention matmuls in FP16 (quant_block.py:630-637), so this is a
floating-point op: the HIP kernel is held to a stated tolerance against the float64 restatement
(oracle.attention_f16) and against a PyTorch fp32 reference at the UNet's full sizes, and must be
no further from them than PyTorch's own FP16 SDPA.  Its fused INT8 output (the operand of
to_out.0) is integer work and is checked BIT-EXACTLY: it must equal quantize() of the kernel's own
FP16 output, through the C-ABI and against the oracle's quantize.

Tolerance (written here, used below): |got - ref| <= 2e-3 + 4e-3 * |ref| — FP16 output rounding
(2^-11 relative) plus the FP16 rounding of P before the second product, for |v| up to a few units.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import detdata as dd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL, RTOL = 2e-3, 4e-3


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def scal(v):
    return torch.tensor(float(v), dtype=torch.float32, device=DEV)


def make(seed, B, tq, tkv, C, fused_layout):
    """q/k/v float16; `fused_layout` puts them in one [B, T, 3C] (self) or k|v in [B, Tkv, 2C]
    buffer so that the kernel reads column slices with a row stride, as in the UNet."""
    if fused_layout and tq == tkv:
        qkv = dd.normal_f16(seed, (B, tq, 3 * C), 1.2)
        return qkv, (qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:])
    q = dd.normal_f16(seed, (B, tq, C), 1.2)
    kv = dd.normal_f16(seed + 1, (B, tkv, 2 * C), 1.2)
    return (q, kv), (q, kv[..., :C], kv[..., C:])


def device_views(host, tq, tkv, C, fused_layout):
    if fused_layout and tq == tkv:
        d = t(host)
        return d[..., :C], d[..., C:2 * C], d[..., 2 * C:]
    q, kv = t(host[0]), t(host[1])
    return q, kv[..., :C], kv[..., C:]


SMALL = [  # B, Tq, Tkv, C, fused layout, forced workgroup shape
    (1, 128, 64, 128, False, 0), (1, 128, 64, 128, False, 2), (1, 128, 64, 128, False, 4),
    (2, 256, 256, 192, True, 0), (2, 256, 256, 192, True, 2), (2, 256, 256, 192, True, 4),
    (1, 100, 77, 128, False, 0),       # ragged queries and keys (cross-attention: 77 tokens)
    (2, 96, 130, 64, False, 2), (2, 96, 130, 64, False, 4),
    (3, 33, 1, 64, False, 0),          # a single key: softmax == 1, out == v
    (1, 1, 300, 128, False, 0),        # a single query
    (1, 64, 640, 64, True, 0),         # 10 key tiles: the prefetch ring wraps
    (1, 320, 320, 128, True, 4),
    # more shapes on the 64-query (two-wave) workgroups
    (2, 256, 256, 192, True, 2), (2, 96, 130, 64, False, 2), (1, 64, 640, 64, True, 2),
    (1, 320, 704, 128, False, 2), (1, 1, 300, 128, False, 2),
    # the short-key kernel (tkv <= 128; automatic there, forced = 1): one and two tiles, whole and ragged,
    # ragged queries, a single key, batch + fused k|v layout
    (1, 128, 64, 128, False, 1), (1, 100, 77, 128, False, 1), (2, 96, 128, 64, False, 1),
    (3, 33, 1, 64, False, 1), (2, 300, 100, 192, False, 1), (2, 128, 128, 128, True, 1),
]


def test_attention_workgroup_shapes_agree_bit_for_bit_and_rows_of_a_batch_equal_single_runs(C):
    """The 128-query and the 64-query workgroups run the same per-wave arithmetic, and the launch
    geometry is chosen per image: an image's result does not depend on the batch it runs in (what
    makes batch sharding over GPUs exact, tests/test_unet_full_gpu.py)."""
    B, tq, tkv, Cc = 3, 320, 704, 128
    host, _ = make(77, B, tq, tkv, Cc, False)
    qd, kd, vd = device_views(host, tq, tkv, Cc, False)
    big = C.attention_f16(qd, kd, vd, Cc // 64, _cfg=4)
    small = C.attention_f16(qd, kd, vd, Cc // 64, _cfg=2)
    auto = C.attention_f16(qd, kd, vd, Cc // 64)
    assert torch.equal(big, small) and torch.equal(big, auto)
    for b in range(B):
        one = C.attention_f16(qd[b:b + 1], kd[b:b + 1], vd[b:b + 1], Cc // 64)
        assert torch.equal(one, auto[b:b + 1])


@pytest.mark.parametrize("B,tq,tkv,Cc", [(2, 1024, 77, 1280), (1, 100, 77, 128), (3, 160, 128, 64), (2, 96, 64, 192),
                                         (1, 4096, 77, 640), (2, 130, 1, 64), (1, 256, 65, 128)])
def test_attention_short_key_kernel_is_bit_identical_to_the_pipelined_one(C, B, tq, tkv, Cc):
    """tkv <= 128 (the UNet's cross-attention: 77 keys) runs attn_short_kernel -- the same arithmetic in the
    same order, unpipelined, at four workgroups per CU.  Same bits as both pipelined launch shapes, FP16 and
    INT8 output; beyond 128 keys forcing it is an error."""
    host, _ = make(91, B, tq, tkv, Cc, False)
    qd, kd, vd = device_views(host, tq, tkv, Cc, False)
    heads = Cc // 64
    auto = C.attention_f16(qd, kd, vd, heads)
    assert torch.equal(auto, C.attention_f16(qd, kd, vd, heads, _cfg=1))
    assert torch.equal(auto, C.attention_f16(qd, kd, vd, heads, _cfg=4))
    assert torch.equal(auto, C.attention_f16(qd, kd, vd, heads, _cfg=2))
    s_inv, zp = scal(float(np.float32(1) / np.float32(0.0173))), scal(7.0)
    assert torch.equal(C.attention_f16(qd, kd, vd, heads, s_inv, zp, _cfg=1),
                       C.attention_f16(qd, kd, vd, heads, s_inv, zp, _cfg=4))
    long_k = t(dd.normal_f16(92, (B, 129, Cc), 1.0))
    with pytest.raises(RuntimeError):
        C.attention_f16(qd, long_k, long_k, heads, _cfg=1)


@pytest.mark.parametrize("B,tq,tkv,Cc,cfg", [(1, 1024, 1024, 1280, 0), (2, 256, 256, 192, 4), (1, 320, 704, 128, 2),
                                             (2, 100, 77, 128, 0)])
def test_attention_prefetch_payload_changes_nothing(C, B, tq, tkv, Cc, cfg):
    """mixdq_attention_f16_prefetch: the payload workgroups only READ the given ranges (the weights of the
    layers behind the attention); the attention result is the plain launch's, bit for bit -- for full and
    ragged ranges, sizes that are not multiples of 16 bytes or of the payload's stripe, empty and missing
    ranges, FP16 and INT8 output; the ranges themselves are untouched; more than 16 ranges are refused."""
    host, _ = make(55, B, tq, tkv, Cc, tq == tkv)
    qd, kd, vd = device_views(host, tq, tkv, Cc, tq == tkv)
    heads = Cc // 64
    want = C.attention_f16(qd, kd, vd, heads, _cfg=cfg)
    g = torch.Generator(device="cpu").manual_seed(5)
    ranges = [torch.randint(-128, 128, (n,), generator=g, dtype=torch.int8).to(DEV)
              for n in (13 * 1024 * 1024 + 7, 4096, 1, 48, 1280 * 1280, 16, 5 * 1024 * 1024 + 16)]
    ranges += [torch.randn(640, 640, generator=g).half().to(DEV)]                     # an FP16 fallback weight
    before = [r.clone() for r in ranges]
    got = C.attention_f16(qd, kd, vd, heads, _cfg=cfg, _prefetch=ranges)
    assert torch.equal(got, want)
    assert all(torch.equal(a, b) for a, b in zip(ranges, before))
    assert torch.equal(C.attention_f16(qd, kd, vd, heads, _cfg=cfg, _prefetch=[None, ranges[0][:0], ranges[4]]), want)
    s_inv, zp = scal(float(np.float32(1) / np.float32(0.0173))), scal(7.0)
    assert torch.equal(C.attention_f16(qd, kd, vd, heads, s_inv, zp, _cfg=cfg, _prefetch=ranges[:3]),
                       C.attention_f16(qd, kd, vd, heads, s_inv, zp, _cfg=cfg))
    with pytest.raises(RuntimeError):
        C.attention_f16(qd, kd, vd, heads, _cfg=cfg, _prefetch=ranges + ranges + ranges[:1])


@pytest.mark.parametrize("case", SMALL, ids=[f"b{c[0]}_q{c[1]}_k{c[2]}_c{c[3]}_{'f' if c[4] else 's'}_w{c[5]}" for c in SMALL])
def test_attention_vs_oracle(C, oracle, case):
    B, tq, tkv, Cc, fused, cfg = case
    host, (q, k, v) = make(31, B, tq, tkv, Cc, fused)
    heads = Cc // 64
    ref16, ref64 = oracle.attention_f16(q, k, v, heads)
    qd, kd, vd = device_views(host, tq, tkv, Cc, fused)
    got = C.attention_f16(qd, kd, vd, heads, _cfg=cfg)
    assert got.shape == (B, tq, Cc) and got.dtype == torch.float16 and got.is_contiguous()
    g = got.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    err = np.abs(g - ref64)
    assert (err <= ATOL + RTOL * np.abs(ref64)).all(), f"max err {err.max():.3e}"
    # and no further from the exact result than PyTorch's own FP16 SDPA on the same inputs
    sd = F.scaled_dot_product_attention(
        qd.unflatten(-1, (heads, 64)).transpose(1, 2), kd.unflatten(-1, (heads, 64)).transpose(1, 2),
        vd.unflatten(-1, (heads, 64)).transpose(1, 2)).transpose(1, 2).reshape(B, tq, Cc)
    err_sd = np.abs(sd.cpu().numpy().astype(np.float64) - ref64)
    assert err.max() <= 1.5 * err_sd.max() + 1e-3
    assert np.sqrt((err ** 2).mean()) <= 1.5 * np.sqrt((err_sd ** 2).mean()) + 1e-5


@pytest.mark.parametrize("case", SMALL, ids=[f"b{c[0]}_q{c[1]}_k{c[2]}_c{c[3]}_{'f' if c[4] else 's'}_w{c[5]}" for c in SMALL])
def test_attention_quantized_output_is_quantize_of_fp16_output(C, oracle, case):
    B, tq, tkv, Cc, fused, cfg = case
    host, _ = make(41, B, tq, tkv, Cc, fused)
    heads = Cc // 64
    qd, kd, vd = device_views(host, tq, tkv, Cc, fused)
    s_inv, zp = float(np.float32(1) / np.float32(0.0173)), 7.0
    o16 = C.attention_f16(qd, kd, vd, heads, _cfg=cfg)
    o8 = C.attention_f16(qd, kd, vd, heads, scal(s_inv), scal(zp), _cfg=cfg)
    assert o8.dtype == torch.int8 and o8.shape == o16.shape
    want = oracle.quantize(o16.cpu().numpy(), s_inv, zp, C.FLAGS & 1)
    assert np.array_equal(o8.cpu().numpy(), want)
    assert torch.equal(o8, C.quantize_per_tensor_to_int8(o16, scal(s_inv), scal(zp)))
    assert (want != want.flat[0]).any()          # the scale actually exercises the range


FULL = [  # the SDXL UNet's attention shapes at 1024 px (B, Tq, Tkv, C)
    (1, 4096, 4096, 640), (1, 1024, 1024, 1280), (2, 4096, 77, 640), (2, 1024, 77, 1280),
]


@pytest.mark.parametrize("case", FULL, ids=[f"b{c[0]}_q{c[1]}_k{c[2]}_c{c[3]}" for c in FULL])
def test_attention_full_size_vs_torch_fp32(C, case):
    B, tq, tkv, Cc = case
    heads = Cc // 64
    g = torch.Generator(device="cpu").manual_seed(5)
    if tq == tkv:
        qkv = (torch.randn(B, tq, 3 * Cc, generator=g) * 1.3).half().to(DEV)
        q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
    else:
        q = (torch.randn(B, tq, Cc, generator=g) * 1.3).half().to(DEV)
        kv = (torch.randn(B, tkv, 2 * Cc, generator=g) * 1.3).half().to(DEV)
        k, v = kv[..., :Cc], kv[..., Cc:]

    def heads_first(x):
        return x.float().unflatten(-1, (heads, 64)).transpose(1, 2)
    s = heads_first(q) @ heads_first(k).transpose(-1, -2) * 0.125
    ref = (s.softmax(-1) @ heads_first(v)).transpose(1, 2).reshape(B, tq, Cc)
    got = C.attention_f16(q, k, v, heads).float()
    err = (got - ref).abs()
    assert bool((err <= ATOL + RTOL * ref.abs()).all()), f"max err {err.max().item():.3e}"
    # size-independent property: rows of softmax sum to 1 => attention of constant v is constant
    ones = torch.ones_like(v)
    c = C.attention_f16(q, k, ones, heads)
    assert torch.equal(c, torch.ones_like(c))


def test_attention_permutation_invariance(C):
    """Permuting the keys (and values with them) must not change the result beyond rounding."""
    B, T, Cc, heads = 1, 256, 128, 2
    g = torch.Generator(device="cpu").manual_seed(9)
    q = torch.randn(B, T, Cc, generator=g).half().to(DEV)
    k = torch.randn(B, T, Cc, generator=g).half().to(DEV)
    v = torch.randn(B, T, Cc, generator=g).half().to(DEV)
    perm = torch.randperm(T, generator=g).to(DEV)
    a = C.attention_f16(q, k, v, heads).float()
    b = C.attention_f16(q, k[:, perm].contiguous(), v[:, perm].contiguous(), heads).float()
    assert bool(((a - b).abs() <= ATOL + RTOL * a.abs()).all())


def test_attention_argument_checks(C):
    q = torch.zeros(1, 64, 128, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError):
        C.attention_f16(q, q, q, 4)                      # head_dim 32
    with pytest.raises(RuntimeError):
        C.attention_f16(q.float(), q, q, 2)
    with pytest.raises(RuntimeError):
        C.attention_f16(q, q[:, :32], q[:, :48], 2)      # k / v disagree
    odd = torch.zeros(1, 64, 132, dtype=torch.float16, device=DEV)[..., :128]
    with pytest.raises(RuntimeError):
        C.attention_f16(odd, odd, odd, 2)                # row stride not a multiple of 8
    empty = C.attention_f16(q[:, :0], q, q, 2)
    assert empty.shape == (1, 0, 128)


def test_attention_xcd_map_changes_where_a_block_runs_not_what_it_computes(C):
    """The XCD-aware workgroup map (csrc/attention.hip attn_block_of, round 6) is a permutation of the workgroup ->
    (batch, head, query block) assignment: with MIXDQ_ATTN_XCD=0 (read once per process: a child process) the pipelined
    kernel (both launch geometries, ragged query / key counts, INT8 output) and the short-key kernel return the bits
    they return with the map on."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [(2, 1024, 1024, 1280), (3, 320, 704, 128), (1, 4096, 4096, 640), (2, 1024, 77, 1280), (1, 1, 300, 128)]
    code = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
import mixdq_amd._C as C
from tests import detdata as dd
import numpy as np
out = []
for B, tq, tkv, Cc in %r:
    q = torch.from_numpy(dd.normal_f16(901, (B, tq, Cc), 1.0)).cuda()
    kv = torch.from_numpy(dd.normal_f16(902, (B, tkv, 2 * Cc), 1.0)).cuda()
    s, z = torch.tensor(30.0, device="cuda"), torch.tensor(2.0, device="cuda")
    for quant in (False, True):
        o = C.attention_f16(q, kv[..., :Cc], kv[..., Cc:], Cc // 64, *((s, z) if quant else ()))
        out.append(hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest())
print("HASHES " + " ".join(out))
''' % (root, shapes)
    got = {}
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, MIXDQ_ATTN_XCD=flag),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        got[flag] = [ln for ln in r.stdout.splitlines() if ln.startswith("HASHES ")][-1]
    assert got["1"] == got["0"] and len(got["1"].split()) == 1 + 2 * len(shapes)
