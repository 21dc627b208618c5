// a2 + a3 (+ a4): INT8 x INT8 -> INT32 GEMM and NHWC implicit-GEMM conv2d for gfx950, with the
// reference's FP32 epilogue  D = f16((f32(acc) - bias0) * scale [+ bias]).
// Replaces qlinear.cc:13-137 + cutlassGemm_*Alignment.cu and qconv2d.cc:27-206 +
// cutlassConv2d_*.cu + conv_act_zero_point_propagate.cu of the reference.
//
// Design (MI355X-first, not a CUTLASS translation):
//   * one kernel family for Linear and Conv2d: both operands are "rows of K-contiguous bytes";
//     the conv's activation row is a gather over (r, s, c) done by the per-lane SOURCE address
//     of the LDS-DMA (global_load_lds_dwordx4); out-of-image taps, the M/N tails and the K tail
//     read a 16-byte zero page instead, so the main loop has no predication.
//   * v_mfma_i32_32x32x32_i8 with the WEIGHT tile as the A operand and the ACTIVATION tile as
//     the B operand: the accumulator then has the output row (m) on the lane and 4 consecutive
//     output channels per register quad, so the per-channel epilogue vectors are float4 loads and
//     the f16 results go to LDS as 8-byte stores and out to HBM as whole 16-byte row segments.
//   * LDS tiles are [rows][BK bytes] with the 16-byte chunk index XOR-swizzled by the row
//     (ds_read_b128 conflict-free); because LDS-DMA writes lane-linear, the swizzle is applied to
//     the source address and to the fragment read (both-sides rule).
//   * STAGES (2-4) LDS buffers, STAGES-1 K-tiles of DMA in flight behind one counted vmcnt wait
//     and one raw s_barrier per K-tile.
//   * padded convs: per-pixel bias0 = zp * (sum of in-bounds taps of wsum) is looked up from a
//     (R*R*S*S) x K table of tap-rectangle sums indexed by the pixel's border class, instead of
//     materialising an [N,P,Q,K] f32 tensor per call as the reference does.
//   * blockIdx -> tile map is XCD-aware (blocks that share a weight panel share an L2).
#include "igemm_kernel.h"
#include "igemm_pp.h"

namespace mixdq {
namespace {

// ---- small-alignment / generic fallback (K % 16 != 0 or C % 16 != 0): one output per thread.
// Replaces the reference's *_smallAlignment CUTLASS instantiations (conv_in C=4, tests K=8).
template <bool CONV>
__global__ __launch_bounds__(256) void igemm_generic_kernel(const IgemmParams p) {
  const int64_t total = p.M * p.N;
  const float zpv = p.table ? *p.zp : 0.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / p.N;
    const int n = (int)(t - m * p.N);
    int acc = 0;
    float b0;
    if constexpr (!CONV) {
      const int8_t* a = p.A + m * p.Ktot;
      const int8_t* w = p.Wt + (int64_t)n * p.Ktot;
      for (int k = 0; k < p.Ktot; ++k) acc += (int)a[k] * (int)w[k];
      b0 = p.bias0[n];
    } else {
      const int pq = p.P * p.Q;
      const int64_t img = m / pq;
      const int rem = (int)(m - img * pq);
      const int pp = rem / p.Q, qq = rem - pp * p.Q;
      const int hb = pp * p.stride - p.pad, wb = qq * p.stride - p.pad;
      for (int r = 0; r < p.R; ++r)
        for (int s = 0; s < p.S; ++s) {
          const int hh = hb + r, ww = wb + s;
          if ((unsigned)hh >= (unsigned)p.H || (unsigned)ww >= (unsigned)p.W) continue;
          const int8_t* a = p.A + ((img * p.H + hh) * p.W + ww) * (int64_t)p.C;
          const int8_t* w = p.Wt + ((int64_t)n * p.R * p.S + r * p.S + s) * p.C;
          for (int c = 0; c < p.C; ++c) acc += (int)a[c] * (int)w[c];
        }
      if (p.table) {
        const int rlo = max(0, -hb), rhi = max(min(p.R - 1, p.H - 1 - hb), 0);
        const int slo = max(0, -wb), shi = max(min(p.S - 1, p.W - 1 - wb), 0);
        const int cls = ((min(rlo, p.R - 1) * p.R + rhi) * p.S + min(slo, p.S - 1)) * p.S + shi;
        b0 = __fmul_rn(p.table[(int64_t)cls * p.N + n], zpv);
      } else {
        b0 = p.bias0[n];
      }
    }
    int64_t drow = m;
    if (p.grp_rows > 0) {
      const int64_t gq = m / p.grp_rows;
      drow = gq * p.grp_stride + p.grp_off + (m - gq * p.grp_rows);
    }
    const bool hb_ = p.bias != nullptr;
    __half o = epilogue_one(acc, b0, p.scale[n], hb_ ? __half2float(p.bias[n]) : 0.f, hb_,
                            p.unfused != 0);
    if (p.res != nullptr)
      o = f32_to_f16_rn(__fadd_rn(__half2float(o), __half2float(p.res[(m / p.res_div) * p.N + n])));
    p.D[drow * p.N + n] = o;
  }
}

// table[cls][k] = sum of wsum[k][r][s] over r in [rlo,rhi], s in [slo,shi] (float adds in (r,s)
// order, exact: integers < 2^24), cls = ((rlo*R + rhi)*S + slo)*S + shi.
__global__ __launch_bounds__(256) void border_table_kernel(const float* __restrict__ wsum,
                                                           float* __restrict__ table, int K,
                                                           int R, int S) {
  const int ncls = R * R * S * S;
  const int64_t total = (int64_t)ncls * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % K);
    int cls = (int)(t / K);
    const int shi = cls % S; cls /= S;
    const int slo = cls % S; cls /= S;
    const int rhi = cls % R;
    const int rlo = cls / R;
    float acc = 0.f;
    for (int r = rlo; r <= rhi; ++r)
      for (int s = slo; s <= shi; ++s) acc += wsum[((int64_t)k * R + r) * S + s];
    table[t] = acc;
  }
}

// a4 stand-alone: the reference's materialised zero-point propagation
// (conv_act_zero_point_propagate.cu:23-51), k fastest for coalesced stores.
__global__ __launch_bounds__(256) void zp_propagate_kernel(const float* __restrict__ wsum,
                                                           const float* __restrict__ zp_p,
                                                           float* __restrict__ out, int N, int H,
                                                           int W, int K, int R, int S, int P,
                                                           int Q, int stride, int pad) {
  const float zp = *zp_p;
  const int64_t total = (int64_t)N * P * Q * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % K);
    int64_t r_ = t / K;
    const int qq = (int)(r_ % Q); r_ /= Q;
    const int pp = (int)(r_ % P);
    const int hb = pp * stride - pad, wb = qq * stride - pad;
    float acc = 0.f;
    for (int r = 0; r < R; ++r)
      for (int s = 0; s < S; ++s) {
        const int hh = hb + r, ww = wb + s;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) acc += wsum[((int64_t)k * R + r) * S + s];
      }
    out[t] = __fmul_rn(acc, zp);
  }
}

// FP16 debug GEMM (qlinear_fp_reference): one output per thread, k-ordered FP32 fmaf chain.
__global__ __launch_bounds__(256) void gemm_f16_kernel(const __half* __restrict__ A,
                                                       const __half* __restrict__ B,
                                                       __half* __restrict__ D, int64_t M, int N,
                                                       int K) {
  const int64_t total = M * N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / N;
    const int n = (int)(t - m * N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k)
      acc = __builtin_fmaf(__half2float(A[m * K + k]), __half2float(B[(int64_t)k * N + n]), acc);
    D[t] = f32_to_f16_rn(acc);
  }
}

// Kernel configurations.  id 0 = automatic choice; ids 1.. can be forced through bits 8..15 of
// the `flags` argument of the C entry points (tuning / tests only).
// X(id, BM, BN, BK, STAGES, WM, WN, KSPLIT, MT, PHASED): block tile, K-tile bytes, LDS stages, wave
// grid (m x n), k-split groups (x WM*WN waves each), MFMA shape (32: 32x32x32, 16: 16x16x64), the
// four-phase main loop
#define MIXDQ_IGEMM_CONFIGS(X)            \
  X(1, 64, 64, 64, 2, 2, 2, 1, 32, false)        \
  X(3, 128, 128, 64, 2, 2, 2, 1, 32, false)      \
  X(4, 64, 64, 128, 3, 2, 2, 1, 32, false)       \
  X(13, 256, 128, 64, 3, 4, 2, 1, 32, false)     \
  X(14, 256, 256, 64, 3, 4, 2, 1, 32, false)     \
  X(15, 128, 256, 64, 3, 2, 4, 1, 32, false)     \
  X(18, 256, 128, 128, 2, 4, 2, 1, 32, false)    \
  X(20, 256, 256, 128, 2, 4, 2, 1, 32, false)    \
  X(25, 128, 320, 128, 2, 4, 2, 1, 32, false)    \
  X(27, 128, 320, 128, 2, 8, 2, 1, 16, false)    \
  X(28, 128, 320, 128, 2, 4, 4, 1, 16, false)    \
  X(35, 128, 128, 64, 3, 4, 2, 1, 32, false)     \
  X(37, 64, 64, 128, 3, 2, 2, 2, 32, false)      \
  X(41, 64, 128, 128, 3, 2, 4, 1, 32, false)     \
  X(42, 64, 80, 128, 3, 4, 1, 2, 16, false)      \
  X(43, 64, 240, 128, 3, 4, 1, 2, 16, false)     \
  X(44, 128, 80, 128, 3, 4, 1, 2, 16, false)     \
  X(45, 64, 80, 128, 4, 4, 1, 2, 16, false)      \
  X(46, 128, 320, 64, 4, 4, 2, 1, 32, false)     \
  X(47, 128, 320, 64, 5, 4, 2, 1, 32, false)     \
  X(56, 64, 80, 128, 6, 4, 1, 2, 16, false)  \
  X(70, 256, 256, 128, 2, 2, 4, 1, 16, true)    \
  X(71, 256, 256, 128, 2, 2, 4, 1, 16, true)
// (71: the persistent form of 70, csrc/igemm_pp.h -- one workgroup per CU walking its tiles; dispatch() runs it where
//  the launch is in its range, and 70's kernel otherwise)

struct TileCfg { int id, bm, bn, bk, stages, wm, wn, ksplit, mt; };
constexpr TileCfg kTileCfgs[] = {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT, PH) {ID, BM, BN, BK, ST, WM, WN, KS, MT},
    MIXDQ_IGEMM_CONFIGS(X)
#undef X
};

// Automatic choice (tools/bench_gemm.py on MI355X): the largest tile that still fills the chip.
// 8-wave 256-row tiles need ~2 blocks per CU of parallelism to pay; long-K problems take the
// 256x256x128 tile (full 128-byte lines per DMA row, fewest L2->LDS bytes per MAC).
// MIXDQ_IGEMM_TUNE="MxNxK=cfg,MxNxK=cfg,...": per-shape overrides of the automatic choice, read
// once (tuning runs: A/B a configuration inside the whole UNet without a rebuild).
struct TuneEntry { int64_t M; int N, K, cfg; };
inline const std::vector<TuneEntry>& tune_overrides() {
  static const std::vector<TuneEntry> table = [] {
    std::vector<TuneEntry> t;
    const char* e = getenv("MIXDQ_IGEMM_TUNE");
    while (e && *e) {
      long long m; int n, k, c, used = 0;
      if (sscanf(e, "%lldx%dx%d=%d%n", &m, &n, &k, &c, &used) == 4) t.push_back({m, n, k, c});
      else { fprintf(stderr, "mixdq: MIXDQ_IGEMM_TUNE: cannot parse \"%s\" (MxNxK=cfg,...); ignored from there\n", e); break; }
      e += used;
      if (*e == ',') ++e;
    }
    return t;
  }();
  return table;
}

// An override applies to the INT8 launches of that shape only (the FP16 layers count K in bytes and
// have their own list), and only if the launch's constraints admit the tile: GEMM+GEGLU needs
// BN % 32 == 0 -- an entry meant for a plain Linear of the same shape is ignored there.
inline int tuned_cfg(int64_t M, int N, int Ktot, bool whole64) {
  for (const TuneEntry& t : tune_overrides()) {
    if (t.M != M || t.N != N || t.K != Ktot) continue;
    for (const TileCfg& c : kTileCfgs)
      if (c.id == t.cfg && (!whole64 || c.bn % 32 == 0)) return t.cfg;
  }
  return 0;
}

// whole64: a GEMM+GEGLU launch (needs BN % 32 == 0: whole value|gate groups per tile; the name is
// round 2's, when the groups were 64 columns)
// phased_ok: a Linear launch on the fast staging path (K % 128 == 0, 32-bit operand offsets)
// tune: consult MIXDQ_IGEMM_TUNE (off for the FP16 layers' rule, which reuses this one on bytes)
inline int select_cfg(int64_t M, int N, int Ktot, bool whole64 = false, bool phased_ok = false,
                      bool tune = true) {
  if (tune) if (const int c = tuned_cfg(M, N, Ktot, whole64)) return c;
  auto blocks = [&](int tm, int tn) {
    return ((M + tm - 1) / tm) * (int64_t)((N + tn - 1) / tn);
  };
  // exactly one (or two) 128x320 workgroups per CU: no tail round, the fewest L2->LDS bytes per MAC
  // among the tiles that still use every CU (M = 8192, N = 1280: 26.0 vs 28.2 us, K = 5120: 66 vs 78)
  const int64_t b320 = blocks(128, 320);
  // (K = 640: its two 128-byte-deep stages are too shallow, 128x128 wins -- except for GEMM+GEGLU,
  // where this tile looks GELU up in LDS: (4096, 5120, 640) 30.3 vs 35.9 us)
  if (N % 320 == 0 && Ktot % 128 == 0 && (Ktot >= 1024 || whole64) &&
      (b320 == kNumCU || b320 == 2 * kNumCU))
    // the 128x320 tile on 16 waves: 8 x 2 waves of 16 x 160 (27; GEMM+GEGLU needs 32-column groups per
    // wave), or -- long K, where the main loop is what counts -- 4 x 4 waves of 32 x 80 (28: 7 fragment
    // reads per 10 MFMAs instead of 11; (8192, 1280, 5120) 53.8 vs 59.6 us, (32768, 640, 2560) 62.6 vs
    // 66.5, (8192, 1280, 1280) 21.2 vs 20.9).  (25: the same tile on 8 waves of 32 x 160.)
    return (!whole64 && Ktot >= 2048) ? 28 : 27;
  // from 1.5 workgroups of 256x256 per CU on: the four-phase loop (fewest L2->LDS bytes per MAC, the
  // reads and the DMA of one wave group under the other's MFMAs): (8192, 10240, 1280) 135 vs 153 us on
  // 256x128, (8192, 3840, 1280) 56 vs 62, (32768, 1920, 640) 77 vs 84 (tools/bench_gemm.py --bs 8)
  // GEMM+GEGLU too, since its epilogue on this tile runs in registers (value and gate of an output sit
  // in the same lane; GELU by table): (8192, 10240, 1280) 117 vs 150 us on 256x128, (32768, 5120, 640)
  // 164 vs 202, (4096, 10240, 1280) 68 vs 79.  (With the staged arithmetic epilogue of round 2 the
  // 256x128 tile won: two workgroups per CU hid part of it.)
  // (short K AND few columns -- the 640-channel to_q / to_out / ff layers at batch >= 8 -- is all
  // prologue and epilogue: three co-resident workgroups of 128x128 pipeline them; (32768, 640, 640):
  // 23.9 us against 28.8 on 256x256, 26.6 on 256x128, 26.3 on 128x320)
  if (!whole64 && Ktot <= 640 && N <= 640 && blocks(128, 128) >= 4 * kNumCU) return 35;
  if (phased_ok && Ktot >= 640 && 2 * blocks(256, 256) >= 3 * kNumCU) return pp_auto(M, N) ? 71 : 70;
  if (blocks(256, 128) >= 2 * kNumCU)
    return (Ktot >= 4096 && blocks(256, 256) >= kNumCU) ? 20 : 13;
  // exact-fit 16x16x64-MFMA tiles (tools/bench_gemm.py, batch 1): 128x80 when that is exactly one
  // or two workgroups per CU and K is long (M = 4096 / 16384 layers with N = 640 / 1280 / 320:
  // 13.5 vs 15.3 us at (4096, 640, 2560); the 3x3 convs at 64x64 and 128x128: 75 vs 91, 43 vs 52 us)
  const int64_t b80 = blocks(128, 80);
  if (!whole64 && N % 80 == 0 && Ktot >= 2048 && (b80 == kNumCU || b80 == 2 * kNumCU)) return 44;
  // ... and 64x80 when THAT is exactly one per CU (M = 1024, N = 1280): every CU streams
  // (64 + 80) * K bytes instead of 160 CUs streaming (64 + 128) * K; four stages for long K
  // ((1024, 1280, 5120): 14.1 vs 17.3 us; 3x3 convs at 32x32: 35.2 vs 36.6), six for K <= 2048
  // where the weights arrive cold from HBM in the UNet (tools/bench_cold.py: 8.0 vs 8.8 us)
  if (!whole64 && N % 80 == 0 && blocks(64, 80) == kNumCU) return Ktot > 2048 ? 45 : 56;
  // 128x128 with 8 waves of 32x64 from ~0.8 workgroups per CU on (measured on the UNet's shapes:
  // tools/bench_gemm.py); below that 64x64 tiles: with 8 waves that split each K-tile's k-steps
  // (cfg 37) when K is long or M tiny -- the chain per K-tile is what bounds these launches --
  // and the plain 4-wave tile (cfg 4) for the short-K GEMMs that put two workgroups on some CUs.
  if (blocks(128, 128) >= 200) return 35;
  if (M <= 256) return 37;
  // 1..2 workgroups of 64x64 per CU (M = 1024, N = 1280): 64x128 tiles with 8 waves of 32x32 leave
  // no second round and halve each wave's chain (7.0 vs 7.5 us, K = 5120: 17.6 vs 18.8)
  if (blocks(64, 64) < 2 * kNumCU && Ktot <= 6144 && N % 128 == 0) return 41;
  return Ktot >= 2048 ? 37 : 4;
}

// Packed-W4 weights run every tile configuration (the weight stage is half the bytes, its pieces
// dealt out wave by wave), but every wave unpacks the fragments it multiplies (6 VALU operations
// per fragment), so tiles whose waves hold few weight fragments per MFMA win: the 32x32x32 tiles
// rather than the 16-row exact-fit ones (tools/bench_gemm.py --w4: (1024, 1280, 5120) 17.1 us on
// the k-split 64x64 tile vs 18.3 on 64x80; (1024, 10240, 1280) 26.7 on 128x128 vs 29.9 on 128x320).
inline int select_cfg_w4(int64_t M, int N, int Ktot, bool whole64 = false) {
  if (const int c = tuned_cfg(M, N, Ktot, whole64)) return c;
  auto blocks = [&](int tm, int tn) {
    return ((M + tm - 1) / tm) * (int64_t)((N + tn - 1) / tn);
  };
  // GEMM+GEGLU (its epilogue runs in registers on every tile, with the GELU table where the tile owns
  // its CU): the 128x320 tile at exactly one or two workgroups per CU, as for W8 ((1024, 10240, 1280):
  // 20.9 vs 25.9 us on 128x128, (4096, 5120, 640): 26.4 vs 30.0); 256x256 for the large launches
  // ((8192, 10240, 1280): 135.5 vs 152 us on 256x128x128; tools/gpu_w4geglu.sh, tools/bench_gemm.py --w4)
  const int64_t b320 = blocks(128, 320);
  if (whole64 && N % 320 == 0 && Ktot % 128 == 0 && (b320 == kNumCU || b320 == 2 * kNumCU)) return 25;
  if (blocks(256, 128) >= 2 * kNumCU) {
    if (whole64) return blocks(256, 256) >= kNumCU ? 20 : 18;
    // plain launches: 256x128x64 (two workgroups per CU) -- (8192, 3840, 1280) 55.1 vs 63.1 us on
    // 256x128x128, (32768, 1920, 640) 65.9 vs 80.6, (8192, 10240, 1280) 140 vs 153
    return (Ktot >= 4096 && blocks(256, 256) >= kNumCU) ? 20 : 13;
  }
  const int64_t b80 = blocks(128, 80);
  if (!whole64 && N % 80 == 0 && Ktot >= 2048 && (b80 == kNumCU || b80 == 2 * kNumCU)) return 44;
  if (blocks(128, 128) >= kNumCU) return 3;
  if (blocks(128, 128) >= 200) return 35;
  if (M <= 256) return 37;
  if (blocks(64, 64) < 2 * kNumCU && Ktot <= 2048 && N % 128 == 0) return 41;
  return Ktot >= 2048 ? 37 : 4;
}

template <bool CONV, bool W4>
int dispatch(IgemmParams& p, hipStream_t stream, int forced_cfg) {
  if (p.M <= 0 || p.N <= 0) return MIXDQ_OK;
  const int align_k = CONV ? p.C : p.Ktot;
  if (align_k % 4 != 0 || p.N % 4 != 0) return MIXDQ_ERR_ALIGNMENT;
  const bool ptr_ok = ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.Wt % 16 == 0) &&
                      ((uintptr_t)p.D % 16 == 0) && ((uintptr_t)p.scale % 16 == 0) &&
                      ((uintptr_t)p.bias0 % 16 == 0) && ((uintptr_t)p.table % 16 == 0) &&
                      ((uintptr_t)p.bias % 8 == 0) && ((uintptr_t)p.res % 16 == 0);
  if constexpr (W4) {
    if (align_k % 32 != 0 || !ptr_ok) return MIXDQ_ERR_W4_SHAPE;   // packed pieces span 32 k
  } else {
    if (align_k % 16 != 0 || !ptr_ok) {
      int64_t blocks = (p.M * p.N + 255) / 256;
      if (blocks > kNumCU * 16) blocks = kNumCU * 16;
      igemm_generic_kernel<CONV><<<(int)blocks, 256, 0, stream>>>(p);
      return launch_status();
    }
  }
  const bool whole64 = p.Dq != nullptr;
  const bool phased_ok = !CONV && p.Ktot % 128 == 0 &&
                         (uint64_t)p.M * (uint64_t)p.Ktot < (1ull << 32) &&
                         (uint64_t)p.N * (uint64_t)p.Ktot < (1ull << 32);
  const int cfg = forced_cfg > 0 ? forced_cfg
                                 : (W4 ? select_cfg_w4(p.M, p.N, p.Ktot, whole64)
                                       : select_cfg(p.M, p.N, p.Ktot, whole64, phased_ok));
  // the four-phase 256 x 256 tile as ONE workgroup per CU walking its tiles (csrc/igemm_pp.h: the next tile's first
  // K-tile lands under the current tile's epilogue; same arithmetic, same bits) wherever a CU has more than one tile
  if constexpr (!CONV && !W4) {
    if (cfg == 71 && phased_ok && pp_ok(p))
      return whole64 ? launch_pp<true>(p, stream) : launch_pp<false>(p, stream);
  }
  switch (cfg) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT, PH) \
  case ID: return launch_tile<BM, BN, BK, ST, WM, WN, CONV, W4, KS, MT, false, PH>(p, stream);
    MIXDQ_IGEMM_CONFIGS(X)
#undef X
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

// to_q GEMM + cross-attention epilogue: the 64x128x128 8-wave tile (two heads per tile), Linear
// fast path only.
template <bool W4>
int launch_att(IgemmParams& p, hipStream_t stream) {
  constexpr int BM = 64, BN = 128, BK = 128, ST = 3;
  constexpr int SMEM = ((igemm_smem_bytes<BM, BN, BK, ST>() + 1023) / 1024) * 1024 + 4 * kStageBytes;
  static_assert(SMEM <= 160 * 1024, "LDS is 160 KiB per CU");
  static bool seen[64] = {};
  if (const int st = lds_opt_in(
          reinterpret_cast<const void*>(&igemm_kernel<BM, BN, BK, ST, 2, 4, false, true, W4, 1, 32, false, true>),
          SMEM, seen))
    return st;
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (p.N + BN - 1) / BN;
  p.gm = tile_map_gm();
  const int64_t grid = (int64_t)p.tiles_m * p.tiles_n;
  if (grid <= 0 || grid > 0x7fffffff || p.tiles_m >= (1 << 24)) return MIXDQ_ERR_INVALID_ARG;
  igemm_kernel<BM, BN, BK, ST, 2, 4, false, true, W4, 1, 32, false, true>
      <<<dim3((unsigned)grid), 512, SMEM, stream>>>(MIXDQ_IGEMM_HEAD_ARGS(p) p);
  return launch_status();
}

// The grouped launch is its own instantiation (GROUPED: the member's operands come from the device
// table, everything else of the argument block stays in scalar registers), built for the tiles a
// set of small independent problems wants.
#define MIXDQ_GROUPED_CONFIGS(X)          \
  X(4, 64, 64, 128, 3, 2, 2, 1, 32)       \
  X(35, 128, 128, 64, 3, 4, 2, 1, 32)     \
  X(37, 64, 64, 128, 3, 2, 2, 2, 32)      \
  X(41, 64, 128, 128, 3, 2, 4, 1, 32)     \
  X(56, 64, 80, 128, 6, 4, 1, 2, 16)

template <bool W4>
int dispatch_grouped(IgemmParams& p, int ngroups, hipStream_t stream, int cfg) {
  p.ngroups_launch = ngroups;
  switch (cfg) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) \
  case ID: return launch_tile<BM, BN, BK, ST, WM, WN, false, W4, KS, MT, false, false, true>(p, stream);
    MIXDQ_GROUPED_CONFIGS(X)
#undef X
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

// ---- FP16 layers (the reference's FP fallback, nn/Linear.py:155-156, nn/Conv2d.py:306-309) ----
// The tile configurations the FP16 instantiation is built for (K in bytes = 2 x elements; the
// automatic choice is the INT8 rule on the byte counts, mapped into this list).
#define MIXDQ_F16_CONFIGS(X)              \
  X(4, 64, 64, 128, 3, 2, 2, 1, 32)       \
  X(13, 256, 128, 64, 3, 4, 2, 1, 32)     \
  X(20, 256, 256, 128, 2, 4, 2, 1, 32)    \
  X(25, 128, 320, 128, 2, 4, 2, 1, 32)    \
  X(35, 128, 128, 64, 3, 4, 2, 1, 32)     \
  X(41, 64, 128, 128, 3, 2, 4, 1, 32)

// Floating-point accumulation is order-sensitive, and an image must not change in its last bits
// with the batch it runs in (the tile rule looks at M = batch x rows): every FP16 configuration
// therefore runs the SAME accumulation -- one MFMA shape (32x32x16), no k-split groups, k ascending
// -- so that any tile of the list gives bit-identical results (INT8 accumulation is exact: free).
inline int select_cfg_f16(int64_t M, int N, int k_bytes) {
  const int c = select_cfg(M, N, k_bytes, false, false, /*tune=*/false);
  switch (c) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) case ID:
    MIXDQ_F16_CONFIGS(X)
#undef X
      return c;
    case 37: return 4;              // k-split 64x64 -> the plain 64x64 tile
    case 45: case 56: case 42: return 41;   // exact-fit 64x80 (16x16 MFMA, k-split) -> 64x128
    case 27: case 28: return 25;    // 16 waves of 16x16x64 MFMAs -> the same tile on 8 waves of 32x32
    case 70: case 71: case 14: case 18: return 20;
    default: return 35;
  }
}

// Small-alignment FP16 fallback (C % 8 != 0: conv_in has 4 input channels): one output per thread,
// FP32 fmaf chain in (r, s, c) order, bias added in FP32, one rounding.
template <bool CONV>
__global__ __launch_bounds__(256) void f16_generic_kernel(const IgemmParams p) {
  const __half* A = reinterpret_cast<const __half*>(p.A);
  const __half* Wt = reinterpret_cast<const __half*>(p.Wt);
  const int C = p.C / 2, K = p.Ktot / 2;        // elements
  const int64_t total = p.M * p.N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = t / p.N;
    const int n = (int)(t - m * p.N);
    float acc = 0.f;
    if constexpr (!CONV) {
      const __half* a = A + m * K;
      const __half* w = Wt + (int64_t)n * K;
      for (int k = 0; k < K; ++k) acc = __builtin_fmaf(__half2float(a[k]), __half2float(w[k]), acc);
    } else {
      const int pq = p.P * p.Q;
      const int64_t img = m / pq;
      const int rem = (int)(m - img * pq);
      const int pp = rem / p.Q, qq = rem - pp * p.Q;
      const int hb = pp * p.stride - p.pad, wb = qq * p.stride - p.pad;
      for (int r = 0; r < p.R; ++r)
        for (int s = 0; s < p.S; ++s) {
          const int hh = hb + r, ww = wb + s;
          if ((unsigned)hh >= (unsigned)p.H || (unsigned)ww >= (unsigned)p.W) continue;
          const __half* a = A + ((img * p.H + hh) * p.W + ww) * (int64_t)C;
          const __half* w = Wt + ((int64_t)n * p.R * p.S + r * p.S + s) * C;
          for (int c = 0; c < C; ++c)
            acc = __builtin_fmaf(__half2float(a[c]), __half2float(w[c]), acc);
        }
    }
    if (p.bias != nullptr) acc = __fadd_rn(acc, __half2float(p.bias[n]));
    __half o = f32_to_f16_rn(acc);
    if (p.res != nullptr)
      o = f32_to_f16_rn(__fadd_rn(__half2float(o), __half2float(p.res[(m / p.res_div) * p.N + n])));
    p.D[m * p.N + n] = o;
  }
}

template <bool CONV>
int dispatch_f16(IgemmParams& p, hipStream_t stream, int forced_cfg) {
  if (p.M <= 0 || p.N <= 0) return MIXDQ_OK;
  const int align_k = CONV ? p.C : p.Ktot;     // bytes
  const bool ptr_ok = ((uintptr_t)p.A % 16 == 0) && ((uintptr_t)p.Wt % 16 == 0) &&
                      ((uintptr_t)p.D % 16 == 0) && ((uintptr_t)p.bias % 8 == 0) &&
                      ((uintptr_t)p.res % 16 == 0);
  if (align_k % 16 != 0 || p.N % 4 != 0 || !ptr_ok) {
    int64_t blocks = (p.M * p.N + 255) / 256;
    if (blocks > kNumCU * 16) blocks = kNumCU * 16;
    f16_generic_kernel<CONV><<<(int)blocks, 256, 0, stream>>>(p);
    return launch_status();
  }
  const int cfg = forced_cfg > 0 ? forced_cfg : select_cfg_f16(p.M, p.N, p.Ktot);
  switch (cfg) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) \
  case ID: return launch_tile<BM, BN, BK, ST, WM, WN, CONV, false, KS, MT, true>(p, stream);
    MIXDQ_F16_CONFIGS(X)
#undef X
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

}  // namespace
}  // namespace mixdq

using namespace mixdq;

extern "C" int mixdq_qlinear_w8a8_rows(const int8_t* A, const int8_t* W, const float* bias0,
                                       const float* scale, const void* bias_f16_or_null,
                                       void* D_f16, int64_t M, int N, int K, int group_rows,
                                       int group_stride, int group_offset,
                                       const void* residual_f16_or_null, int64_t residual_row_div,
                                       int flags, mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A || !W || !bias0 || !scale || !D_f16) return MIXDQ_ERR_INVALID_ARG;
  IgemmParams p{};
  p.A = A; p.Wt = W; p.bias0 = bias0; p.scale = scale; p.bias = (const __half*)bias_f16_or_null;
  p.table = nullptr; p.zp = nullptr; p.D = (__half*)D_f16;
  p.M = M; p.N = N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.grp_rows = group_rows; p.grp_stride = group_stride; p.grp_off = group_offset;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  if (p.res && group_rows > 0) return MIXDQ_ERR_ROWMAP_RESIDUAL;   // residual rows follow m, not D_row
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  if (flags & MIXDQ_FLAG_W4) return dispatch<false, true>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
  return dispatch<false, false>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
}

extern "C" int mixdq_qlinear_w8a8_geglu(const int8_t* A, const int8_t* W, const float* bias0,
                                        const float* scale, const void* bias_f16_or_null,
                                        int8_t* out_i8, int64_t M, int N, int K,
                                        const float* out_scale_inv, const float* out_zero_point,
                                        int flags, mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A || !W || !bias0 || !scale || !out_i8 || !out_scale_inv || !out_zero_point)
    return MIXDQ_ERR_INVALID_ARG;
  // whole value/gate groups per tile, the LDS-DMA kernels only, 8-byte output stores
  if (N % 32 != 0 || K % 16 != 0 || ((uintptr_t)out_i8 & 7)) return MIXDQ_ERR_GEGLU_SHAPE;
  if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)scale | (uintptr_t)bias0) & 15)
    return MIXDQ_ERR_GEGLU_SHAPE;
  if ((uintptr_t)bias_f16_or_null & 7) return MIXDQ_ERR_GEGLU_SHAPE;
  IgemmParams p{};
  p.A = A; p.Wt = W; p.bias0 = bias0; p.scale = scale; p.bias = (const __half*)bias_f16_or_null;
  if (const int st = ensure_gelu_table((hipStream_t)stream)) return st;
  p.D = nullptr; p.Dq = out_i8; p.g_sinv = out_scale_inv; p.g_zp = out_zero_point;
  p.M = M; p.N = N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.res_div = 1;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  if (flags & MIXDQ_FLAG_W4) return dispatch<false, true>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
  return dispatch<false, false>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
}

extern "C" int mixdq_qlinear_w8a8_grouped(const int8_t* A, const mixdq_gemm_group* groups_device,
                                          int ngroups, int64_t M, int max_N, int K, int group_rows,
                                          int group_stride, int group_offset, int flags,
                                          mixdq_stream_t stream) {
  if (M < 0 || max_N < 0 || K < 0 || ngroups < 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || max_N == 0 || ngroups == 0) return MIXDQ_OK;
  if (!A || !groups_device || ngroups > 65535) return MIXDQ_ERR_INVALID_ARG;
  const bool w4 = flags & MIXDQ_FLAG_W4;
  if (K % (w4 ? 32 : 16) != 0 || max_N % 4 != 0 || ((uintptr_t)A & 15))
    return w4 ? MIXDQ_ERR_W4_SHAPE : MIXDQ_ERR_ALIGNMENT;   // the LDS-DMA kernels only
  IgemmParams p{};
  p.A = A; p.groups = groups_device;
  p.M = M; p.N = max_N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.grp_rows = group_rows; p.grp_stride = group_stride; p.grp_off = group_offset;
  p.res_div = 1;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  int cfg = (flags >> 8) & 0xff;
  // the members are independent problems: the tile only has to suit one of them, the grid
  // (x ngroups) fills the chip.  M <= 64: 64x64 k-split tiles, else 128x128 (8 waves).
  if (cfg == 0) cfg = M <= 64 ? 37 : 35;
  return w4 ? dispatch_grouped<true>(p, ngroups, (hipStream_t)stream, cfg)
            : dispatch_grouped<false>(p, ngroups, (hipStream_t)stream, cfg);
}

extern "C" int mixdq_qlinear_w8a8_attn(const int8_t* A, const int8_t* W, const float* bias0,
                                       const float* scale, const void* k_f16, const void* v_f16,
                                       void* out, int64_t M, int N, int K, int rows_per_image,
                                       int tkv, int64_t k_batch_stride, int k_row_stride,
                                       int64_t v_batch_stride, int v_row_stride,
                                       float softmax_scale, const float* out_scale_inv_or_null,
                                       const float* out_zero_point_or_null, int flags,
                                       mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || rows_per_image <= 0 || tkv <= 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A || !W || !bias0 || !scale || !k_f16 || !v_f16 || !out) return MIXDQ_ERR_INVALID_ARG;
  if ((out_scale_inv_or_null == nullptr) != (out_zero_point_or_null == nullptr))
    return MIXDQ_ERR_INVALID_ARG;
  const bool w4 = flags & MIXDQ_FLAG_W4;
  // whole head pairs per tile, whole 64-row tiles per image, at most two key tiles, fast staging
  if (N % 128 != 0 || K % 128 != 0 || rows_per_image % 64 != 0 || M % rows_per_image != 0 ||
      tkv > 2 * kKeys || (uint64_t)M * (uint64_t)K >= (1ull << 32) ||
      (uint64_t)N * (uint64_t)K >= (1ull << 32))
    return MIXDQ_ERR_SHAPE;
  if (k_row_stride % 8 || v_row_stride % 8 || k_batch_stride % 8 || v_batch_stride % 8 ||
      (((uintptr_t)A | (uintptr_t)W | (uintptr_t)k_f16 | (uintptr_t)v_f16 | (uintptr_t)bias0 |
        (uintptr_t)scale) & 15) || ((uintptr_t)out & (out_scale_inv_or_null ? 7 : 15)))
    return MIXDQ_ERR_ALIGNMENT;
  IgemmParams p{};
  p.A = A; p.Wt = W; p.bias0 = bias0; p.scale = scale;
  p.M = M; p.N = N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.res_div = 1;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  p.att_k = (const __half*)k_f16; p.att_v = (const __half*)v_f16;
  p.att_k_bs = k_batch_stride; p.att_v_bs = v_batch_stride;
  p.att_k_rs = k_row_stride; p.att_v_rs = v_row_stride;
  p.att_tkv = tkv; p.att_tq = rows_per_image;
  p.att_scale_log2 = softmax_scale * 1.4426950408889634f;
  p.att_out = out; p.att_sinv = out_scale_inv_or_null; p.att_zp = out_zero_point_or_null;
  return w4 ? launch_att<true>(p, (hipStream_t)stream) : launch_att<false>(p, (hipStream_t)stream);
}

extern "C" int mixdq_qlinear_w8a8(const int8_t* A, const int8_t* W, const float* bias0,
                                  const float* scale, const void* bias_f16_or_null, void* D_f16,
                                  int64_t M, int N, int K, int flags, mixdq_stream_t stream) {
  return mixdq_qlinear_w8a8_rows(A, W, bias0, scale, bias_f16_or_null, D_f16, M, N, K, 0, 0, 0,
                                 nullptr, 1, flags, stream);
}

extern "C" size_t mixdq_qconv2d_workspace_bytes(int K, int R, int S, int pad) {
  if (pad <= 0 || K <= 0 || R <= 0 || S <= 0) return 0;
  return (size_t)R * R * S * S * K * sizeof(float);
}

extern "C" int mixdq_conv_border_table(const float* wsum_krs, float* table, int K, int R, int S,
                                       mixdq_stream_t stream) {
  if (!wsum_krs || !table || K <= 0 || R <= 0 || S <= 0) return MIXDQ_ERR_INVALID_ARG;
  const int64_t total = (int64_t)R * R * S * S * K;
  int64_t blocks = (total + 255) / 256;
  if (blocks > kNumCU * 8) blocks = kNumCU * 8;
  border_table_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(wsum_krs, table, K, R, S);
  return launch_status();
}

extern "C" int mixdq_qconv2d_w8a8_table(const int8_t* X, const int8_t* Wt, const float* scale,
                                        const float* table_or_null, const float* zero_point,
                                        const float* bias0_or_null, const void* bias_f16_or_null,
                                        void* D, int N, int H, int W, int C, int K, int R, int S,
                                        int stride, int pad, const void* residual_f16_or_null,
                                        int64_t residual_row_div, int flags,
                                        mixdq_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
    return MIXDQ_ERR_INVALID_ARG;
  if (!X || !Wt || !scale || !D) return MIXDQ_ERR_INVALID_ARG;
  if (pad > 0 ? (!table_or_null || !zero_point) : !bias0_or_null) return MIXDQ_ERR_INVALID_ARG;
  // every output pixel's window must overlap the image (border classes are non-empty rectangles)
  if (pad >= R || pad >= S) return MIXDQ_ERR_PADDING;
  const int P = (H + 2 * pad - (R - 1) - 1) / stride + 1;
  const int Q = (W + 2 * pad - (S - 1) - 1) / stride + 1;
  if (P <= 0 || Q <= 0 || N == 0) return MIXDQ_OK;
  // 3x3 / stride 1 / pad 1 with the input halo resident in LDS (csrc/iconv.hip): the automatic choice
  // wherever it applies; tile ids 90 .. 93 force it, any other forced id keeps the implicit-GEMM family
  // (MIXDQ_HALO_CONV=0: off, for A/B runs)
  {
    static const bool halo_on = [] { const char* e = getenv("MIXDQ_HALO_CONV"); return !(e && e[0] == '0'); }();
    const int forced = (flags >> 8) & 0xff;
    const bool aligned = !(((uintptr_t)X | (uintptr_t)Wt | (uintptr_t)scale | (uintptr_t)table_or_null |
                            (uintptr_t)D | (uintptr_t)residual_f16_or_null) & 15) &&
                         !((uintptr_t)bias_f16_or_null & 7);
    int tile = 0;
    if (!(flags & MIXDQ_FLAG_W4) && aligned && ((forced >= 90 && forced <= 93) || (forced == 0 && halo_on)))
      tile = halo_conv_select(N, H, W, C, K, R, S, stride, pad);
    if (forced >= 90 && forced <= 93) {
      if (tile == 0 || (forced != 91 && W % 16 != 0) || (forced >= 92 && H % 16 != 0)) return MIXDQ_ERR_SHAPE;
      tile = forced;
    }
    if ((flags & MIXDQ_FLAG_UPSAMPLE2X) && (tile == 0 || (H & 1) || (W & 1))) return MIXDQ_ERR_SHAPE;
    if (tile != 0) {
      HaloConvArgs a{};
      a.ups = (flags & MIXDQ_FLAG_UPSAMPLE2X) ? 1 : 0;
      a.X = X; a.Wt = Wt; a.scale = scale; a.bias = (const __half*)bias_f16_or_null;
      a.table = table_or_null; a.zp = zero_point; a.D = (__half*)D;
      a.res = (const __half*)residual_f16_or_null;
      a.res_div = residual_row_div > 0 ? residual_row_div : 1;
      if (a.res && a.res_div != 1 && a.res_div != (int64_t)H * W) return MIXDQ_ERR_INVALID_ARG;
      a.NI = N; a.H = H; a.W = W; a.C = C; a.K = K;
      a.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
      return halo_conv_launch(a, tile, (hipStream_t)stream);
    }
  }
  IgemmParams p{};
  p.A = X; p.Wt = Wt; p.scale = scale; p.bias = (const __half*)bias_f16_or_null;
  p.bias0 = pad > 0 ? nullptr : bias0_or_null;
  p.table = pad > 0 ? table_or_null : nullptr;
  p.zp = zero_point; p.D = (__half*)D;
  p.M = (int64_t)N * P * Q; p.N = K; p.Ktot = R * S * C;
  p.H = H; p.W = W; p.C = C; p.R = R; p.S = S; p.P = P; p.Q = Q; p.stride = stride; p.pad = pad;
  p.grp_rows = 0; p.grp_stride = 0; p.grp_off = 0;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  if (flags & MIXDQ_FLAG_W4) return dispatch<true, true>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
  return dispatch<true, false>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
}

extern "C" int mixdq_qconv2d_w8a8(const int8_t* X, const int8_t* Wt, const float* scale,
                                  const float* wsum_krs_or_null, const float* zero_point,
                                  const float* bias0_or_null, const void* bias_f16_or_null,
                                  void* D, void* workspace, int N, int H, int W, int C, int K,
                                  int R, int S, int stride, int pad, int dilation, int flags,
                                  mixdq_stream_t stream) {
  if (dilation != 1) return MIXDQ_ERR_UNSUPPORTED;
  const float* table = nullptr;
  if (pad > 0) {
    if (!wsum_krs_or_null || !workspace) return MIXDQ_ERR_INVALID_ARG;
    int st = mixdq_conv_border_table(wsum_krs_or_null, (float*)workspace, K, R, S, stream);
    if (st != MIXDQ_OK) return st;
    table = (const float*)workspace;
  }
  return mixdq_qconv2d_w8a8_table(X, Wt, scale, table, zero_point, bias0_or_null,
                                  bias_f16_or_null, D, N, H, W, C, K, R, S, stride, pad, nullptr, 1,
                                  flags, stream);
}

extern "C" int mixdq_conv_zero_point_propagate(const float* wsum_krs, const float* zero_point,
                                               float* out, int N, int H, int W, int K, int R,
                                               int S, int stride, int pad,
                                               mixdq_stream_t stream) {
  if (!wsum_krs || !zero_point || !out) return MIXDQ_ERR_INVALID_ARG;
  const int P = (H + 2 * pad - (R - 1) - 1) / stride + 1;
  const int Q = (W + 2 * pad - (S - 1) - 1) / stride + 1;
  const int64_t total = (int64_t)N * P * Q * K;
  if (total <= 0) return MIXDQ_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > kNumCU * 8) blocks = kNumCU * 8;
  zp_propagate_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(wsum_krs, zero_point, out, N,
                                                                     H, W, K, R, S, P, Q, stride,
                                                                     pad);
  return launch_status();
}

extern "C" int mixdq_gemm_f16(const void* A, const void* B, void* D, int64_t M, int N, int K,
                              mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A || !B || !D) return MIXDQ_ERR_INVALID_ARG;
  int64_t blocks = (M * N + 255) / 256;
  if (blocks > kNumCU * 16) blocks = kNumCU * 16;
  gemm_f16_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>((const __half*)A, (const __half*)B,
                                                                (__half*)D, M, N, K);
  return launch_status();
}

extern "C" int mixdq_linear_f16(const void* A_f16, const void* W_f16, const void* bias_f16_or_null,
                                void* D_f16, int64_t M, int N, int K,
                                const void* residual_f16_or_null, int64_t residual_row_div,
                                int flags, mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A_f16 || !W_f16 || !D_f16) return MIXDQ_ERR_INVALID_ARG;
  if ((int64_t)K * 2 > 0x7fffffff) return MIXDQ_ERR_SHAPE;
  IgemmParams p{};
  p.A = (const int8_t*)A_f16; p.Wt = (const int8_t*)W_f16; p.bias = (const __half*)bias_f16_or_null;
  p.D = (__half*)D_f16;
  p.M = M; p.N = N; p.Ktot = 2 * K;           // the kernel family counts K in bytes
  p.H = p.W = p.P = p.Q = 1; p.C = 2 * K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  return dispatch_f16<false>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
}

extern "C" int mixdq_conv2d_f16(const void* X_f16, const void* Wt_f16, const void* bias_f16_or_null,
                                void* D_f16, int N, int H, int W, int C, int K, int R, int S,
                                int stride, int pad, const void* residual_f16_or_null,
                                int64_t residual_row_div, int flags, mixdq_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || R <= 0 || S <= 0 || stride <= 0 || pad < 0)
    return MIXDQ_ERR_INVALID_ARG;
  if (!X_f16 || !Wt_f16 || !D_f16) return MIXDQ_ERR_INVALID_ARG;
  const int P = (H + 2 * pad - (R - 1) - 1) / stride + 1;
  const int Q = (W + 2 * pad - (S - 1) - 1) / stride + 1;
  if (P <= 0 || Q <= 0 || N == 0) return MIXDQ_OK;
  IgemmParams p{};
  p.A = (const int8_t*)X_f16; p.Wt = (const int8_t*)Wt_f16; p.bias = (const __half*)bias_f16_or_null;
  p.D = (__half*)D_f16;
  p.M = (int64_t)N * P * Q; p.N = K; p.Ktot = R * S * C * 2;
  p.H = H; p.W = W; p.C = 2 * C; p.R = R; p.S = S; p.P = P; p.Q = Q; p.stride = stride; p.pad = pad;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  return dispatch_f16<true>(p, (hipStream_t)stream, (flags >> 8) & 0xff);
}

extern "C" const char* mixdq_status_string(int status) {
  switch (status) {
    case MIXDQ_OK: return "ok";
    case MIXDQ_ERR_INVALID_ARG: return "invalid argument (null pointer or bad size)";
    case MIXDQ_ERR_ALIGNMENT:
      return "Int8 kernel with input or output alignment not to 4 is not supported.";
    case MIXDQ_ERR_UNSUPPORTED: return "unsupported configuration (dilation must be 1)";
    case MIXDQ_ERR_LAUNCH: return "HIP kernel launch failed";
    case MIXDQ_ERR_W4_SHAPE:
      return "unsupported configuration (packed 4-bit weights need K % 32 == 0 -- conv: C % 32 == 0 "
             "-- and 16-byte aligned operands)";
    case MIXDQ_ERR_GEGLU_SHAPE:
      return "unsupported configuration (GEMM+GEGLU needs N % 32 == 0, K % 16 == 0, 16-byte aligned "
             "operands and an 8-byte aligned output)";
    case MIXDQ_ERR_PADDING:
      return "unsupported configuration (padding must be smaller than the kernel size)";
    case MIXDQ_ERR_ROWMAP_RESIDUAL:
      return "unsupported configuration (an output row map and a residual cannot be combined)";
    case MIXDQ_ERR_SHAPE:
      return "unsupported configuration (shape outside this fused kernel's range)";
    default: return "unknown status";
  }
}

extern "C" int mixdq_abi_version(void) { return MIXDQ_ABI_VERSION; }

extern "C" int mixdq_gelu_table(uint16_t* out_device, mixdq_stream_t stream) {
  if (!out_device) return MIXDQ_ERR_INVALID_ARG;
  if (const int st = ensure_gelu_table((hipStream_t)stream)) return st;
  return hipMemcpyFromSymbolAsync(out_device, HIP_SYMBOL(g_gelu_tab), kGeluTabBytes, 0,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess
             ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}

#if MIXDQ_STAMP
// diagnostic builds only: register (or clear, with null) the stamp buffer, [grid][16 waves][16] uint64
extern "C" int mixdq_debug_stamps(void* buffer) {
  unsigned long long b = (unsigned long long)(uintptr_t)buffer;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &b, sizeof(b)) == hipSuccess ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}
#endif

// a Linear problem (k_align == k_total: no taps) that takes the fast staging path
static bool linear_fast(int64_t M, int N, int k_align, int k_total) {
  return k_align == k_total && k_total % 128 == 0 &&
         (uint64_t)M * (uint64_t)k_total < (1ull << 32) && (uint64_t)N * (uint64_t)k_total < (1ull << 32);
}

extern "C" int mixdq_igemm_select_id(int64_t M, int N, int k_align, int k_total) {
  if (M <= 0 || N <= 0 || k_align % 4 != 0 || N % 4 != 0) return -1;
  if (k_align % 16 != 0) return 0;   // generic kernel
  return select_cfg(M, N, k_total, false, linear_fast(M, N, k_align, k_total));
}

extern "C" int mixdq_conv_halo_select(int N, int H, int W, int C, int K, int R, int S, int stride,
                                      int pad) {
  const char* e = getenv("MIXDQ_HALO_CONV");
  if (e && e[0] == '0') return 0;
  return halo_conv_select(N, H, W, C, K, R, S, stride, pad);
}

extern "C" int mixdq_igemm_select_id_w4(int64_t M, int N, int k_align, int k_total) {
  if (M <= 0 || N <= 0 || k_align % 32 != 0 || N % 4 != 0) return -1;
  return select_cfg_w4(M, N, k_total);
}

extern "C" int mixdq_igemm_select_id_geglu(int64_t M, int N, int k_total, int w4) {
  if (M <= 0 || N <= 0 || N % 32 != 0 || k_total % (w4 ? 32 : 16) != 0) return -1;
  return w4 ? select_cfg_w4(M, N, k_total, true)
            : select_cfg(M, N, k_total, true, linear_fast(M, N, k_total, k_total));
}

extern "C" int mixdq_igemm_select(int64_t M, int N, int k_align, int k_total, int* bm, int* bn,
                                  int* bk, int* stages) {
  if (!bm || !bn || !bk || !stages || M <= 0 || N <= 0) return MIXDQ_ERR_INVALID_ARG;
  if (k_align % 4 != 0 || N % 4 != 0) return MIXDQ_ERR_ALIGNMENT;
  if (k_align % 16 != 0) { *bm = *bn = *bk = *stages = 0; return MIXDQ_OK; }   // generic kernel
  const int cfg = select_cfg(M, N, k_total, false, linear_fast(M, N, k_align, k_total));
  for (const TileCfg& c : kTileCfgs)
    if (c.id == cfg) { *bm = c.bm; *bn = c.bn; *bk = c.bk; *stages = c.stages; }
  return MIXDQ_OK;
}
