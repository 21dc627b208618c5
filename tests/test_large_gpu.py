"""Parity at the sizes of BASELINE.json configs[3] / configs[4] (batch 8 per GPU of the bs-64 run,
UNet batch 16 = batch 8 x classifier-free guidance): the HIP path runs the FULL problem
(M = 8192 .. 16384 per 1024-token level, M = 131072 .. 262144 pixels at level 0) and is checked

  * bit for bit against the oracle on a seeded SAMPLE of output rows -- a GEMM row depends only on
    its own input row, so the oracle's result for the sampled rows is the full problem's;
  * for convolutions, bit for bit against the same kernel run image by image (batch
    equivariance), with one whole image pinned to the oracle;
  * through size-independent properties (linearity of the accumulator, permutation equivariance).

These are the launches that take the large tiles (256x128 / 256x256 / 128x320) -- the batch-1
suites above never select them automatically."""
import numpy as np
import pytest
import torch

from tests import detdata as dd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def scal(v):
    return torch.tensor(float(v), dtype=torch.float32, device=DEV)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint16) if a.dtype == np.float16 else a


def sample_rows(M, n, seed):
    rng = np.random.RandomState(seed)
    rows = np.unique(np.concatenate([[0, 1, M // 2, M - 2, M - 1], rng.randint(0, M, n)]))
    return rows


LIN = [  # batch, tokens per image, N, K, bias   (SURVEY.md Appendix A shapes)
    (8, 1024, 1280, 1280, True), (16, 1024, 1280, 1280, False), (8, 1024, 3840, 1280, False),
    (8, 1024, 1280, 5120, True), (8, 4096, 640, 640, True), (16, 4096, 1920, 640, False),
    (8, 4096, 640, 2560, True), (16, 1024, 10240, 1280, True), (8, 76, 2560, 2048, False),
]


@pytest.mark.parametrize("B,T,N,K,bias", LIN, ids=[f"b{c[0]}_t{c[1]}_n{c[2]}_k{c[3]}" for c in LIN])
def test_qlinear_full_batch_sampled_rows_vs_oracle(C, oracle, B, T, N, K, bias):
    M = B * T
    g = torch.Generator(device="cpu").manual_seed(1000 + M + N)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
    w = dd.int8(2, (N, K))
    b0, sc = dd.f32(3, (N,), -4000, 4000), dd.f32(4, (N,), 1e-4, 1e-3)
    bs = dd.f16(5, (N,), -1, 1) if bias else None
    res = (torch.randn(M, N, generator=g) * 0.5).half()
    out = C.qlinear_w8_a8_ohalf(a.to(DEV), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0),
                                None if bs is None else t(bs))
    out_res = C.qlinear_w8_a8_ohalf(a.to(DEV), t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0),
                                    None if bs is None else t(bs), _residual=res.to(DEV))
    rows = sample_rows(M, 160, M + N + K)
    want = oracle.qlinear(a.numpy()[rows], w, b0, sc, bs, C.FLAGS & 1)
    got = out.cpu().numpy()[rows]
    assert np.array_equal(bits(got), bits(want)), f"{(bits(got) != bits(want)).sum()} differ"
    want_res = oracle.add_f16(want, res.numpy()[rows])
    assert np.array_equal(bits(out_res.cpu().numpy()[rows]), bits(want_res))
    # which tile ran: at these sizes the automatic choice must be one of the large / 8-wave tiles
    assert C.igemm_select_id(M, N, K) in (13, 20, 25, 27, 28, 35, 41, 44, 70, 71), C.igemm_select_id(M, N, K)


@pytest.mark.parametrize("B", [8, 16])
def test_qlinear_geglu_full_batch_sampled_rows_vs_oracle(C, oracle, B):
    """ff.net.0.proj + GEGLU + quantize at (B x 1024, 10240, 1280): the largest launch of the step."""
    M, D, K = B * 1024, 5120, 1280
    g = torch.Generator(device="cpu").manual_seed(77 + B)
    a = torch.randint(-128, 128, (M, K), generator=g, dtype=torch.int8)
    w = dd.int8(52, (2 * D, K))
    scale, bias0 = dd.f32(53, (2 * D,), 2e-5, 9e-5), dd.f32(54, (2 * D,), -3000, 3000)
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    perm = C.geglu_row_order(D, DEV)
    got = C.qlinear_geglu(a.to(DEV), t(w)[perm].contiguous(), t(scale)[perm].contiguous(),
                          t(bias0)[perm].contiguous(), None, scal(s_inv), scal(zp))
    rows = sample_rows(M, 96, 5 * B)
    h = oracle.qlinear(a.numpy()[rows], w, bias0, scale, None, C.FLAGS & 1)
    q_ref, _ = oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)
    assert np.array_equal(got.cpu().numpy()[rows], q_ref)
    assert len(np.unique(q_ref)) > 32


CONV = [  # batch, H = W, Cin, Cout, ksize, stride
    (8, 32, 1280, 1280, 3, 1), (16, 32, 1280, 1280, 3, 1), (8, 64, 640, 640, 3, 1),
    (8, 128, 320, 320, 3, 1), (8, 128, 320, 320, 3, 2), (8, 64, 1280, 640, 1, 1),
]


@pytest.mark.parametrize("B,HW,Cin,Cout,ks,stride", CONV,
                         ids=[f"b{c[0]}_{c[1]}px_c{c[2]}_k{c[3]}_{c[4]}x{c[4]}_s{c[5]}" for c in CONV])
def test_qconv2d_full_batch_equals_per_image_and_oracle(C, oracle, B, HW, Cin, Cout, ks, stride):
    pad = ks // 2
    g = torch.Generator(device="cpu").manual_seed(31 * B + HW)
    x = torch.randint(-128, 128, (B, HW, HW, Cin), generator=g, dtype=torch.int8)   # NHWC
    wt = dd.int8(61, (Cout, ks, ks, Cin))
    sc = dd.f32(62, (Cout,), 1e-5, 1e-4)
    bias = dd.f16(63, (Cout,), -1, 1)
    zp = -37.0
    wsum = wt.astype(np.float32).sum(axis=3, dtype=np.float32)
    b0 = (wsum.reshape(Cout, -1).sum(axis=1, dtype=np.float32) * np.float32(zp)).astype(np.float32)

    def run(xi):
        return C.qconv2d_w8_a8_ohalf(
            xi.to(DEV).permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2), t(sc), scal(1), scal(zp),
            t(sc), t(wsum.reshape(Cout, 1, ks, ks)) if pad else None, None if pad else t(b0),
            t(bias), stride, pad)

    full = run(x)
    assert full.shape[0] == B
    for i in (0, B // 2, B - 1):                         # batch equivariance, bit for bit
        assert torch.equal(full[i:i + 1], run(x[i:i + 1])), f"image {i}"
    # one image against the oracle (a top band of rows keeps the C loop short)
    rows = 6
    band = x[B - 1:B, :rows + pad].numpy()               # rows 0 .. rows+pad-1 feed outputs 0 .. rows-1
    want = oracle.qconv2d(band, wt, sc, wsum if pad else None, zp, None if pad else b0, bias,
                          stride, pad, C.FLAGS & 1)
    n_out = (rows + stride - 1) // stride if stride > 1 else rows
    n_out = min(n_out, want.shape[1] - (1 if pad else 0))   # the band's last row lacks its lower tap
    got = full[B - 1].permute(1, 2, 0)[:n_out].contiguous().cpu().numpy()
    assert np.array_equal(bits(got), bits(want[0, :n_out])), "band vs oracle"


def test_large_m_linearity_and_permutation(C):
    """(16384, 1280, 1280) -- UNet batch 16: out(a1) + out(a2) == out(a1 + a2) in the integer
    domain (scale 1, bias0 0, values small enough that f16 is exact) and row-permutation
    equivariance, on the 256-row tiles."""
    M, K, N = 16384, 1280, 1280
    g = torch.Generator(device="cpu").manual_seed(5)
    a1 = torch.randint(-1, 2, (M, K), generator=g, dtype=torch.int8).to(DEV)
    a2 = torch.randint(-1, 2, (M, K), generator=g, dtype=torch.int8).to(DEV)
    w = torch.randint(-1, 2, (N, K), generator=g, dtype=torch.int8).to(DEV)
    one, zero = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)

    def f(a):
        return C.qlinear_w8_a8_ohalf(a, w, one, scal(1), scal(0), zero, one, zero, None)

    o1, o2, o12 = f(a1), f(a2), f(a1 + a2)
    assert torch.equal(o1.float() + o2.float(), o12.float())
    assert torch.equal(o1.float(), a1.float() @ w.float().t())
    perm = torch.randperm(M, generator=g).to(DEV)
    assert torch.equal(f(a1[perm]), o1[perm])


@pytest.mark.parametrize("B", [8, 16])
def test_producers_full_batch_sampled_rows_vs_oracle(C, oracle, B):
    """LayerNorm+quantize and GEGLU+quantize at B x 1024 rows, attention at B images: sampled rows /
    images against the oracle."""
    M, Cc = B * 1024, 1280
    x = (dd.normal_f16(31, (M, Cc), 1.2).astype(np.float32) + 0.4).astype(np.float16)
    gamma = (dd.normal_f16(32, (Cc,), 0.3).astype(np.float32) + 1).astype(np.float16)
    beta = dd.normal_f16(33, (Cc,), 0.2)
    qp = [(float(np.float32(1) / np.float32(0.02)), -7.0)]
    outs, _ = C.layernorm_quantize(t(x), t(gamma), t(beta), 1e-5, [(scal(a), scal(b)) for a, b in qp])
    rows = sample_rows(M, 64, B)
    o_ref, _ = oracle.layernorm_quantize(x[rows], gamma, beta, 1e-5, qp, C.FLAGS & 1)
    assert np.array_equal(outs[0].cpu().numpy()[rows], o_ref[0])
    # attention: B images x 1024 tokens x 20 heads; image B-1 against the float64 oracle
    qkv = dd.normal_f16(41, (B, 1024, 3 * 128), 1.0)
    d = t(qkv)
    att = C.attention_f16(d[..., :128], d[..., 128:256], d[..., 256:], 2)
    _, ref = oracle.attention_f16(qkv[B - 1:, :, :128], qkv[B - 1:, :, 128:256], qkv[B - 1:, :, 256:], 2)
    err = np.abs(att[B - 1:].cpu().numpy().astype(np.float64) - ref)
    assert (err <= 2e-3 + 4e-3 * np.abs(ref)).all(), err.max()
    one = C.attention_f16(d[B - 1:, :, :128], d[B - 1:, :, 128:256], d[B - 1:, :, 256:], 2)
    assert torch.equal(att[B - 1:], one)


# ---- per-GPU batch 32 / 64: the N = 2 and N = 1 shards of BASELINE.json configs[3] (global batch 64) ----
# Inputs are generated on the GPU (hundreds of MB per tensor); only the sampled rows travel to the host.

def _rand_i8(shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randint(-128, 128, shape, generator=g, dtype=torch.int8, device=DEV)


@pytest.mark.parametrize("B", [32, 64])
def test_shard_size_geglu_sampled_rows_vs_oracle(C, oracle, B):
    """ff.net.0.proj + GEGLU + quantize at (B x 1024, 10240, 1280) on the four-phase 256x256 tile."""
    M, D, K = B * 1024, 5120, 1280
    a = _rand_i8((M, K), 900 + B)
    w = dd.int8(52, (2 * D, K))
    scale, bias0 = dd.f32(53, (2 * D,), 2e-5, 9e-5), dd.f32(54, (2 * D,), -3000, 3000)
    s_inv, zp = float(np.float32(1) / np.float32(0.02)), -60.0
    perm = C.geglu_row_order(D, DEV)
    got = C.qlinear_geglu(a, t(w)[perm].contiguous(), t(scale)[perm].contiguous(),
                          t(bias0)[perm].contiguous(), None, scal(s_inv), scal(zp))
    assert C.igemm_select_id(M, 2 * D, K, geglu=True) == 71      # the persistent four-phase kernel
    rows = sample_rows(M, 96, 7 * B)
    idx = torch.from_numpy(rows).to(DEV)
    h = oracle.qlinear(a[idx].cpu().numpy(), w, bias0, scale, None, C.FLAGS & 1)
    q_ref, _ = oracle.geglu_quantize(h, s_inv, zp, C.FLAGS & 1)
    assert np.array_equal(got[idx].cpu().numpy(), q_ref)
    # batch equivariance: three images alone (their own tile rule: M = 1024 runs on 128x320)
    for i in (0, B // 2, B - 1):
        one = C.qlinear_geglu(a[i * 1024:(i + 1) * 1024], t(w)[perm].contiguous(),
                              t(scale)[perm].contiguous(), t(bias0)[perm].contiguous(), None,
                              scal(s_inv), scal(zp))
        assert torch.equal(got[i * 1024:(i + 1) * 1024], one), f"image {i}"


@pytest.mark.parametrize("B", [32, 64])
def test_shard_size_linear_640_sampled_rows_vs_oracle(C, oracle, B):
    """(B x 4096, 640, 640) with a residual epilogue: the 640-channel to_q / to_out layers (M = 262144
    rows at batch 64)."""
    M, N, K = B * 4096, 640, 640
    a = _rand_i8((M, K), 300 + B)
    w = dd.int8(2, (N, K))
    b0, sc, bs = dd.f32(3, (N,), -4000, 4000), dd.f32(4, (N,), 1e-4, 1e-3), dd.f16(5, (N,), -1, 1)
    g = torch.Generator(device=DEV).manual_seed(5 + B)
    res = (torch.randn(M, N, generator=g, device=DEV) * 0.5).half()
    out = C.qlinear_w8_a8_ohalf(a, t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), t(bs),
                                _residual=res)
    assert C.igemm_select_id(M, N, K) == 35
    rows = sample_rows(M, 160, M + N)
    idx = torch.from_numpy(rows).to(DEV)
    want = oracle.add_f16(oracle.qlinear(a[idx].cpu().numpy(), w, b0, sc, bs, C.FLAGS & 1),
                          res[idx].cpu().numpy())
    assert np.array_equal(bits(out[idx].cpu().numpy()), bits(want))
    for i in (0, B // 2, B - 1):
        sl = slice(i * 4096, (i + 1) * 4096)
        one = C.qlinear_w8_a8_ohalf(a[sl], t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), t(bs),
                                    _residual=res[sl].contiguous())
        assert torch.equal(out[sl], one), f"image {i}"


@pytest.mark.parametrize("B", [32, 64])
@pytest.mark.parametrize("ks,Cin,Cout", [(3, 320, 320), (1, 640, 320)],
                         ids=["3x3_halo_128px_c320", "1x1_shortcut_128px_640to320"])
def test_shard_size_conv_equals_per_image_and_oracle(C, oracle, B, ks, Cin, Cout):
    """Level-0 convs at B x 128 x 128 pixels (1 048 576 output pixels at batch 64): the halo-resident 3x3
    kernel and the 1x1 shortcut (second half of a split: bias0 form + residual epilogue)."""
    HW, pad = 128, ks // 2
    x = _rand_i8((B, HW, HW, Cin), 31 * B + ks)                  # NHWC memory
    wt = dd.int8(61, (Cout, ks, ks, Cin))
    sc = dd.f32(62, (Cout,), 1e-5, 1e-4)
    bias = dd.f16(63, (Cout,), -1, 1)
    zp = -37.0
    wsum = wt.astype(np.float32).sum(axis=3, dtype=np.float32)
    b0 = (wsum.reshape(Cout, -1).sum(axis=1, dtype=np.float32) * np.float32(zp)).astype(np.float32)
    g = torch.Generator(device=DEV).manual_seed(11 + B)
    res = None
    if ks == 1:                                                  # the other half's output, added in the epilogue
        res = (torch.randn(B, HW, HW, Cout, generator=g, device=DEV) * 0.5).half().permute(0, 3, 1, 2)

    def run(xi, ri):
        return C.qconv2d_w8_a8_ohalf(
            xi.permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2), t(sc), scal(1), scal(zp), t(sc),
            t(wsum.reshape(Cout, 1, ks, ks)) if pad else None, None if pad else t(b0), t(bias), 1, pad,
            _residual=ri)

    full = run(x, res)
    if ks == 3:
        assert C.conv_halo_select(B, HW, HW, Cin, Cout, 3, 3, 1, 1) == 93      # 16 x 16 pixels x 160 channels
    for i in (0, B // 2, B - 1):                                 # batch equivariance, bit for bit
        assert torch.equal(full[i:i + 1], run(x[i:i + 1], None if res is None else res[i:i + 1])), f"image {i}"
    rows = 4                                                     # a top band of the LAST image vs the oracle
    band = x[B - 1:B, :rows + pad].cpu().numpy()
    want = oracle.qconv2d(band, wt, sc, wsum if pad else None, zp, None if pad else b0, bias, 1, pad,
                          C.FLAGS & 1)
    got = full[B - 1].permute(1, 2, 0)[:rows].contiguous().cpu().numpy()
    want = want[0, :rows]
    if res is not None:
        want = oracle.add_f16(want.reshape(-1, Cout),
                              res[B - 1].permute(1, 2, 0)[:rows].reshape(-1, Cout).cpu().numpy()
                              ).reshape(want.shape)
    assert np.array_equal(bits(got), bits(want)), "band vs oracle"


def test_qlinear_operand_over_4_gib_leaves_the_32bit_offset_path(C, oracle):
    """M x K >= 2^32 bytes: the fast staging path keeps one 32-bit byte offset per lane and is gated on
    the operand fitting it (csrc/igemm.hip launch_tile); above that the launch must take the general
    path with 64-bit row pointers -- nothing at the UNet's sizes reaches the gate."""
    M, N, K = 2_200_000, 64, 2048                                # 4.5 GB of int8 activations
    a = _rand_i8((M, K), 4242)
    w = dd.int8(2, (N, K))
    b0, sc = dd.f32(3, (N,), -4000, 4000), dd.f32(4, (N,), 1e-4, 1e-3)
    out = C.qlinear_w8_a8_ohalf(a, t(w), t(sc), scal(1), scal(0), t(b0), t(sc), t(b0), None)
    rows = np.unique(np.concatenate([sample_rows(M, 64, 1), [2_097_151, 2_097_152, 2_097_153]]))
    idx = torch.from_numpy(rows).to(DEV)                         # rows either side of the 2^32-byte line
    want = oracle.qlinear(a[idx].cpu().numpy(), w, b0, sc, None, C.FLAGS & 1)
    assert np.array_equal(bits(out[idx].cpu().numpy()), bits(want))
