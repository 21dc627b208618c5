// Producer fusions: the FP16 normalisation that feeds a quantized layer,
// fused with that layer's activation quantizer, so the INT8 tensor is produced in one pass instead
// of  norm -> (silu) -> [layout copy] -> quantize.
//
//   GroupNorm (+SiLU) + quantize, NHWC   -> ResnetBlock2D.norm1/norm2 -> conv1/conv2,
//                                           Transformer2DModel.norm -> proj_in
//   LayerNorm + quantize (up to 3 scales) -> BasicTransformerBlock.norm1/2/3 -> to_q,k,v / ff
//   GEGLU + quantize                      -> ff.net.0 -> ff.net.2
//
// The reference fuses none of this (stock PyTorch FP16 ops, then its quantize kernel); the fused
// kernels keep the same rounding points: normalised value -> FP16, SiLU/GELU -> FP16, product ->
// FP16, then the a1 quantizer on that FP16 value.  All reductions have a FIXED order, and the
// transcendental steps are the shared specification include/mixdq_math.h, so the CPU oracle
// (oracle/mixdq_oracle.c) reproduces every kernel bit-for-bit.  HBM-bound.
#include <cstdlib>
#include "common.h"
#include "../../include/mixdq_math.h"

namespace mixdq {
namespace {

struct alignas(16) Half8 { uint32_t w[4]; };
struct alignas(8) Char8 { uint32_t w[2]; };

__device__ __forceinline__ float half_at(const Half8& h, int j) {
  const uint32_t w = h.w[j >> 1];
  __half_raw r;
  r.x = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
  return __half2float(__half(r));
}

__device__ __forceinline__ float round_f16(float v) { return __half2float(f32_to_f16_rn(v)); }

__device__ __forceinline__ void put_half(Half8& h, int j, float v) {
  const uint32_t b = __half_as_ushort(f32_to_f16_rn(v));
  if (j & 1) h.w[j >> 1] = (h.w[j >> 1] & 0x0000ffffu) | (b << 16);
  else h.w[j >> 1] = (h.w[j >> 1] & 0xffff0000u) | b;
}

__device__ __forceinline__ void put_q(Char8& c, int j, int q) {
  const int sh = 8 * (j & 3);
  c.w[j >> 2] = (c.w[j >> 2] & ~(0xffu << sh)) | ((uint32_t)(q & 0xff) << sh);
}

// ------------------------------------------------------------------------------- GroupNorm
// Thread geometry shared by stats and apply (and restated in the oracle):
//   OC = C / 8 channel-octets per pixel, PP = max(1, 256 / OC) pixels per block-iteration,
//   blockDim = OC * PP; thread t owns octet o = t % OC of pixel lane pp = t / OC.
//   An octet spans at most two groups (cg = C / G >= 4): elements j < jb belong to group g0,
//   the rest to g0 + 1.
struct GnGeom {
  int C, G, cg, OC, PP;
  int64_t HW;
  int ppb;      // pixels per block (multiple of PP)
  int nchunk;   // blocks per image
  int ppb_apply, nchunk_apply;   // apply pass split (elementwise: order-free; = the statistics')
  int OCs, PPa, GS;              // apply pass: octets per channel slice (blockIdx.z), its pixel lanes, groups per
                                 // slice (OC, PP, G when the pass is not sliced)
  int C1;       // two-source input (a skip connection never concatenated in memory): channels
                // 0 .. C1-1 come from x [N, HW, C1], the rest from x2 [N, HW, C - C1]; C1 == C: one
};

// this thread's channel octet: source row pointer of pixel 0 of image n and the row stride
__device__ __forceinline__ const __half* gn_src(const __half* x, const __half* x2, const GnGeom& g,
                                                int n, int o, int& stride) {
  if (8 * o < g.C1) { stride = g.C1; return x + ((int64_t)n * g.HW) * g.C1 + 8 * o; }
  stride = g.C - g.C1;
  return x2 + ((int64_t)n * g.HW) * stride + (8 * o - g.C1);
}

// Optional second product of the apply pass: the INPUT itself (no normalisation) quantized per
// source tensor, in that source's own [N, HW, Cs] layout -- the operand of a layer that reads the
// same activation as the norm (the ResNet block's 1x1 shortcut, nn/Conv2d.py:330-343 quantizes each
// half of a split input with its own quantizer).  The apply pass already holds x in registers.
struct GnRaw {
  const float* s_inv[2];
  const float* zp[2];
  int8_t* q[2];
};

template <int U>
__global__ void gn_stats_kernel(const __half* __restrict__ x, const __half* __restrict__ x2,
                                float2* __restrict__ partial, GnGeom g) {
  MIXDQ_ARGS_NOW(x, x2, partial, g.C, g.G, g.cg, g.OC, g.PP, g.HW, g.ppb, g.nchunk, g.C1);
  extern __shared__ float lds[];   // [blockDim][4]: s0, q0, s1, q1
  const int t = threadIdx.x;
  const int o = t % g.OC, pp = t / g.OC;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int g0 = (8 * o) / g.cg;
  const int jb = min(8, (g0 + 1) * g.cg - 8 * o);
  const int64_t p_begin = (int64_t)chunk * g.ppb;
  const int64_t p_end = min(g.HW, p_begin + g.ppb);
  int xs;
  const __half* base = gn_src(x, x2, g, n, o, xs);
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
  // U pixels of this thread in flight at a time (4; 2 where a thread has no more than two; 1 = the plain loop, for A/B
  // runs): as `load; use` per pixel the compiler waits for each load before it issues the next, and a thread's 2 .. 16
  // pixels become as many trips to the memory side (the tensor was written by the producing conv on other XCDs) with
  // two waves per SIMD to hide them.  Unconditional, clamped addresses (no branch between the loads); the additions
  // are the same, in the same order (pixels ascending, j ascending).
  for (int64_t p = p_begin + pp; p < p_end; p += (int64_t)U * g.PP) {
    Half8 h[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      h[u] = *reinterpret_cast<const Half8*>(base + min(p + (int64_t)u * g.PP, p_end - 1) * xs);
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (p + (int64_t)u * g.PP < p_end) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = half_at(h[u], j);
          if (j < jb) { s0 = __fadd_rn(s0, v); q0 = __builtin_fmaf(v, v, q0); }
          else        { s1 = __fadd_rn(s1, v); q1 = __builtin_fmaf(v, v, q1); }
        }
      }
  }
  lds[4 * t + 0] = s0; lds[4 * t + 1] = q0; lds[4 * t + 2] = s1; lds[4 * t + 3] = q1;
  __syncthreads();
  if (t < g.G) {   // group t: octets olo..ohi, pixel lanes ascending, fixed order
    const int olo = (t * g.cg) / 8, ohi = ((t + 1) * g.cg - 1) / 8;
    float s = 0.f, q = 0.f;
    for (int l = 0; l < g.PP; ++l)
      for (int oo = olo; oo <= ohi; ++oo) {
        const int tt = l * g.OC + oo;
        const int first = (8 * oo) / g.cg;          // group of that octet's leading part
        const int part = (first == t) ? 0 : 2;      // leading or trailing accumulator
        if (first == t || first + 1 == t) {
          s = __fadd_rn(s, lds[4 * tt + part]);
          q = __fadd_rn(q, lds[4 * tt + part + 1]);
        }
      }
    partial[((int64_t)n * g.nchunk + chunk) * g.G + t] = make_float2(s, q);
  }
}

// One 64-lane wave per (group, image): lane l sums the chunk partials l, l + 64, ... in order,
// then a 6-step xor butterfly (every lane ends with the same bits).  Fixed order.
__device__ __forceinline__ float wave_sum_gn(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
  return v;
}

// mean / rstd of group `grp` of image n from the chunk partials, by one 64-lane wave; the result
// goes to out[grp] (the apply kernel's LDS table).
__device__ __forceinline__ void gn_finalize_tail(float s, float q, float2* out, const GnGeom& g, float eps,
                                                 int grp, int lane) {
  s = wave_sum_gn(s);
  q = wave_sum_gn(q);
  if (lane == 0) {
    const float cnt = (float)((double)g.HW * g.cg);
    const float mean = s / cnt;
    float var = __builtin_fmaf(-mean, mean, q / cnt);
    var = fmaxf(var, 0.f);
    const float rstd = 1.0f / sqrtf(__fadd_rn(var, eps));
    out[grp] = make_float2(mean, rstd);
  }
}

__device__ __forceinline__ void gn_finalize_group(const float2* __restrict__ partial, float2* out,
                                                  const GnGeom& g, float eps, int grp, int n,
                                                  int lane) {
  float s = 0.f, q = 0.f;
  if (g.nchunk <= 512) {
    // (make_gn_geom: about 512 blocks per image, never more.)  Every partial of this lane is requested before the
    // first is used: written as a loop, the compiler waits for each load before it issues the next -- eight trips
    // to the memory side (the partials come from other XCDs' statistics blocks) in a kernel that is nothing else:
    // 4.8 us per launch, 46 launches per step.  Same additions in the same order (k ascending).
    constexpr int R = 8;
    float2 v[R];
    const float2* base = partial + (int64_t)n * g.nchunk * g.G + grp;
#pragma unroll
    for (int k = 0; k < R; ++k)     // unconditional and clamped (an address that exists): no branch between the loads
      v[k] = base[(int64_t)min(lane + 64 * k, g.nchunk - 1) * g.G];
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (lane + 64 * k < g.nchunk) {
        s = __fadd_rn(s, v[k].x);
        q = __fadd_rn(q, v[k].y);
      }
  } else {
    for (int c = lane; c < g.nchunk; c += 64) {
      const float2 v = partial[((int64_t)n * g.nchunk + c) * g.G + grp];
      s = __fadd_rn(s, v.x);
      q = __fadd_rn(q, v.y);
    }
  }
  gn_finalize_tail(s, q, out, g, eps, grp, lane);
}

// The same, for TWO groups at once and with every partial requested before the first is used (at most
// 512 partials per group: 8 per lane): the apply blocks of the sliced pass run this in their prologue, where
// the loads' latency -- the partials come from other XCDs' statistics blocks, i.e. from the memory side --
// is the whole cost.  Same order of additions as gn_finalize_group (lane l: partials l, l + 64, ... ascending).
__device__ __forceinline__ void gn_finalize_pair(const float2* __restrict__ partial, float2* out,
                                                 const GnGeom& g, float eps, int ga, int gb, bool has_b,
                                                 int n, int lane) {
  constexpr int R = 8;
  float2 va[R], vb[R];
  const float2* base = partial + (int64_t)n * g.nchunk * g.G;
#pragma unroll
  for (int k = 0; k < R; ++k) {   // unconditional (a branch per round made the compiler drain the loads round by
    const int c = min(lane + 64 * k, g.nchunk - 1);   // round); clamped: an address that exists, an L2 hit
    va[k] = base[(int64_t)c * g.G + ga];
    vb[k] = base[(int64_t)c * g.G + gb];
  }
  float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k)
    if (lane + 64 * k < g.nchunk) {
      sa = __fadd_rn(sa, va[k].x); qa = __fadd_rn(qa, va[k].y);
      sb = __fadd_rn(sb, vb[k].x); qb = __fadd_rn(qb, vb[k].y);
    }
  gn_finalize_tail(sa, qa, out, g, eps, ga, lane);
  if (has_b) gn_finalize_tail(sb, qb, out, g, eps, gb, lane);
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float2* __restrict__ partial,
                                                         float2* __restrict__ stats, GnGeom g,
                                                         float eps) {
  gn_finalize_group(partial, stats + (int64_t)blockIdx.y * g.G, g, eps, blockIdx.x, blockIdx.y,
                    threadIdx.x);
}

// ---- SiLU by table (round 6) -----------------------------------------------------------------------------------
// The apply pass rounds the normalised value to FP16, applies SiLU and rounds to FP16 again: SiLU there is a function
// of 16 bits.  Its arithmetic (include/mixdq_math.h: an exp and a correctly rounded division, ~32 of the pass's ~45
// vector operations per element) makes the pass VALU-bound at batch >= 8 ((8, 128 x 128, 320): 75 us with SiLU, 45
// without).  TAB instantiations look f16(silu(y)) up in LDS instead: the table covers the non-trivial range of the
// specification -- y in [0, 8.06) (beyond: silu(y) rounds to y) and (-20.5, 0] (beyond: -0) -- 75 KB, built ONCE per
// device BY the specification (silu_table_init_kernel), copied into LDS by LDS-DMA at block entry by blocks that then
// walk several chunks of their image; a value outside the table (or NaN / inf) takes the arithmetic itself, decided
// per wave.  Bit-identical by construction; used where a launch is large enough to pay for the table (below).
constexpr int kSiluPos = 0x4810, kSiluNeg = 0x4d20;          // magnitudes (FP16 bit patterns) covered per sign
constexpr int kSiluTabBytes = (2 * (kSiluPos + kSiluNeg) + 1023) / 1024 * 1024;
__device__ uint16_t g_silu_tab[kSiluTabBytes / 2];

__global__ __launch_bounds__(256) void silu_table_init_kernel() {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kSiluTabBytes / 2) return;
  unsigned short bits = 0;
  if (i < kSiluPos) bits = (unsigned short)i;
  else if (i < kSiluPos + kSiluNeg) bits = (unsigned short)(0x8000 | (i - kSiluPos));
  __half_raw r;
  r.x = bits;
  g_silu_tab[i] = __half_as_ushort(f32_to_f16_rn(mixdq_siluf(__half2float(__half(r)))));
}

// Built once per device by the first launch that needs it; inside a stream capture the (idempotent) init kernel is
// recorded in front of the first consumer of that capture only (the logic of igemm.hip's ensure_gelu_table).
inline int ensure_silu_table(hipStream_t stream) {
  static bool done[64] = {};
  static unsigned long long in_capture[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
  if (done[dev]) return MIXDQ_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  unsigned long long cap_id = 0;
  if (hipStreamGetCaptureInfo(stream, &cap, &cap_id) != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap != hipStreamCaptureStatusNone && cap_id != 0 && in_capture[dev] == cap_id) return MIXDQ_OK;
  silu_table_init_kernel<<<(kSiluTabBytes / 2 + 255) / 256, 256, 0, stream>>>();
  if (hipGetLastError() != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap == hipStreamCaptureStatusNone) {
    if (hipStreamSynchronize(stream) != hipSuccess) return MIXDQ_ERR_LAUNCH;
    done[dev] = true;
  } else {
    in_capture[dev] = cap_id;
  }
  return MIXDQ_OK;
}

// SELF: the blocks reduce their groups' partials themselves (stats == nullptr); a template parameter because as a
// run-time branch the two forms shared registers and the compiler drained every load where they met.
// TAB: SiLU by the LDS table above; the block walks the chunks blockIdx.x, blockIdx.x + gridDim.x, ... of its image.
template <bool SILU, bool UNFUSED, bool SELF, bool TAB = false>
__global__ void gn_apply_kernel(const __half* __restrict__ x, const __half* __restrict__ x2,
                                const float2* __restrict__ partial,
                                const float2* __restrict__ stats, float eps,
                                const __half* __restrict__ gamma,
                                const __half* __restrict__ beta,
                                const float* __restrict__ s_inv_p, const float* __restrict__ zp_p,
                                int8_t* __restrict__ out_q, __half* __restrict__ out_h, GnGeom g,
                                GnRaw raw) {
  MIXDQ_ARGS_NOW(x, x2, partial, stats, eps, gamma, beta, s_inv_p, zp_p, out_q, out_h);
  MIXDQ_ARGS_NOW(g.C, g.G, g.cg, g.OCs, g.PPa, g.GS, g.HW, g.ppb_apply, g.nchunk, g.C1, raw.s_inv[0],
                 raw.s_inv[1], raw.zp[0], raw.zp[1], raw.q[0], raw.q[1]);
  static_assert(!TAB || (SILU && !SELF), "the table variant: SiLU, statistics from the finalize launch");
  __shared__ float2 s_stats[SELF ? 1024 : 1];   // G <= OC * PP <= 1024
  extern __shared__ __attribute__((aligned(16))) char gn_dyn[];   // TAB: the SiLU table
  if constexpr (TAB) {     // requested first of all, by the block's complete waves, 1 KiB per wave-instruction
    const int nfull = (int)blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < nfull)
      for (int q = wave; q < kSiluTabBytes / 1024; q += nfull)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g_silu_tab) + q * 1024 +
                                                            ((int)threadIdx.x & 63) * 16),
            (__attribute__((address_space(3))) void*)(gn_dyn + q * 1024), 16, 0, 0);
  }
  // the quantizers' scalars as scalar loads at kernel entry (kernel-uniform addresses; as plain loads they were
  // vector loads issued after the statistics, the raw pair behind a pointer fetched from the argument block)
  using cf32 = const __attribute__((address_space(4))) float;
  const bool want_q = out_q != nullptr;
  const float s_inv = want_q ? *(cf32*)s_inv_p : 0.f, zp = want_q ? *(cf32*)zp_p : 0.f;
  const float raw_si0 = raw.q[0] ? *(cf32*)raw.s_inv[0] : 0.f, raw_zp0 = raw.q[0] ? *(cf32*)raw.zp[0] : 0.f;
  const float raw_si1 = raw.q[1] ? *(cf32*)raw.s_inv[1] : 0.f, raw_zp1 = raw.q[1] ? *(cf32*)raw.zp[1] : 0.f;
  const int t = threadIdx.x;
  // channel slice blockIdx.z: octets [z * OCs, (z + 1) * OCs) = groups [z * GS, (z + 1) * GS) -- a slice ends
  // on a group boundary; one slice (OCs = OC, PPa = PP, GS = G) unless the launch sliced the pass
  const int o = (int)blockIdx.z * g.OCs + t % g.OCs, pp = t / g.OCs;
  const int n = blockIdx.y;
  int chunk = blockIdx.x;
  const int g0 = (8 * o) / g.cg;
  const int jb = min(8, (g0 + 1) * g.cg - 8 * o);
  const Half8 gm = *reinterpret_cast<const Half8*>(gamma + 8 * o);
  const Half8 bt = *reinterpret_cast<const Half8*>(beta + 8 * o);
  // the block's first pixel is requested here, beside gamma / beta / the statistics, not behind them: at batch 1
  // a thread has one or two pixels, and the pass was two memory round trips in a row (statistics, then pixels)
  int64_t p_begin = (int64_t)chunk * g.ppb_apply;
  int64_t p_end = min(g.HW, p_begin + g.ppb_apply);
  int xs;
  const __half* src = gn_src(x, x2, g, n, o, xs);
  Half8 h_next;
  h_next.w[0] = h_next.w[1] = h_next.w[2] = h_next.w[3] = 0;
  if (p_begin + pp < p_end) h_next = *reinterpret_cast<const Half8*>(src + (p_begin + pp) * xs);
  float2 st0, st1;
  if constexpr (SELF) {
    // Every block reduces the partials of ITS groups itself (lane l: partials l, l + 64, ...; then the fixed
    // butterfly) -- cheaper than the kernel boundary of a separate finalize launch: all G groups where an
    // image has at most 64 partials per group, and in the sliced pass (round 5) the GS <= 8 groups of the
    // block's channel slice whatever their count; otherwise the finalize kernel ran first (stats != null)
    const int nfull = (int)blockDim.x >> 6, wave = t >> 6;   // complete waves only
    const int gfirst = (int)blockIdx.z * g.GS;
    if (g.GS <= 2 * nfull) {   // the sliced pass: one pair per wave, straight-line (in a loop the compiler drains
      if (wave < nfull && wave < g.GS) {   // the gamma / beta loads before the first partial is requested)
        const bool has_b = wave + nfull < g.GS;
        gn_finalize_pair(partial, s_stats, g, eps, gfirst + wave, gfirst + (has_b ? wave + nfull : wave), has_b,
                         n, t & 63);
      }
    } else if (wave < nfull) {
      for (int r = wave; r < g.GS; r += 2 * nfull) {
        const bool has_b = r + nfull < g.GS;
        gn_finalize_pair(partial, s_stats, g, eps, gfirst + r, gfirst + (has_b ? r + nfull : r), has_b, n,
                         t & 63);
      }
    }
    __syncthreads();
    st0 = s_stats[g0];
    st1 = s_stats[min(g0 + 1, g.G - 1)];
  } else {
    st0 = stats[(int64_t)n * g.G + g0];
    st1 = stats[(int64_t)n * g.G + min(g0 + 1, g.G - 1)];
  }
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float mean = j < jb ? st0.x : st1.x, rstd = j < jb ? st0.y : st1.y;
    a[j] = __fmul_rn(rstd, half_at(gm, j));
    b[j] = __builtin_fmaf(-mean, a[j], half_at(bt, j));
  }
  const int64_t img = ((int64_t)n * g.HW) * g.C + 8 * o;
  const int which = 8 * o < g.C1 ? 0 : 1;
  int8_t* raw_q = which ? raw.q[1] : raw.q[0];
  const float raw_si = which ? raw_si1 : raw_si0, raw_zp = which ? raw_zp1 : raw_zp0;
  if (raw_q) raw_q += ((int64_t)n * g.HW) * xs + (8 * o - (which ? g.C1 : 0));
  if constexpr (TAB) {     // the table has landed, for every wave (the first pixel with it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const uint16_t* tab = reinterpret_cast<const uint16_t*>(gn_dyn);
  for (;;) {               // (one chunk per block unless TAB)
  for (int64_t p = p_begin + pp; p < p_end; p += g.PPa) {
    const Half8 h = h_next;
    if (p + g.PPa < p_end) h_next = *reinterpret_cast<const Half8*>(src + (p + g.PPa) * xs);
    Half8 oh;
    oh.w[0] = oh.w[1] = oh.w[2] = oh.w[3] = 0;
    float y8[8], h8[8];
    if constexpr (TAB) {
      float pre[8];
      uint32_t tb[8];
      bool far = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        h8[j] = half_at(h, j);
        const __half yh = f32_to_f16_rn(__builtin_fmaf(h8[j], a[j], b[j]));     // GroupNorm -> fp16
        pre[j] = __half2float(yh);
        const uint32_t yb = __half_as_ushort(yh), mag = yb & 0x7fffu;
        const bool neg = (yb >> 15) != 0;
        const bool in = mag < (uint32_t)(neg ? kSiluNeg : kSiluPos);
        far |= !in;
        tb[j] = tab[in ? (neg ? kSiluPos + mag : mag) : 0u];                      // f16(silu(y)), from the table
      }
      if (__builtin_amdgcn_ballot_w64(far) != 0) {      // wave-uniform, rare: a value the table does not cover
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t yb = __half_as_ushort(__float2half_rn(pre[j])), mag = yb & 0x7fffu;
          if (mag >= (uint32_t)((yb >> 15) ? kSiluNeg : kSiluPos))
            tb[j] = __half_as_ushort(f32_to_f16_rn(mixdq_siluf(pre[j])));       // the specification itself
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __half_raw r;
        r.x = (unsigned short)tb[j];
        y8[j] = __half2float(__half(r));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) oh.w[j] = tb[2 * j] | (tb[2 * j + 1] << 16);
    } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h8[j] = half_at(h, j);
      float y = round_f16(__builtin_fmaf(h8[j], a[j], b[j]));           // GroupNorm -> fp16
      if (SILU) y = round_f16(mixdq_siluf(y));                          // SiLU -> fp16
      y8[j] = y;
      put_half(oh, j, y);
    }
    }
    // (the quantizers only where their output exists -- kernel-uniform branches -- and with the packed helper:
    //  the pass is VALU-bound, and the raw-input quantizer ran for every element whether or not it was wanted)
    if (want_q) *reinterpret_cast<uint2*>(out_q + img + p * g.C) = quantize_pack8<UNFUSED>(y8, s_inv, zp);
    if (out_h) *reinterpret_cast<Half8*>(out_h + img + p * g.C) = oh;
    if (raw_q) *reinterpret_cast<uint2*>(raw_q + p * xs) = quantize_pack8<UNFUSED>(h8, raw_si, raw_zp);
  }
    if constexpr (!TAB) break;
    chunk += (int)gridDim.x;                       // TAB: the block's next chunk of this image
    if (chunk >= g.nchunk_apply) break;
    p_begin = (int64_t)chunk * g.ppb_apply;
    p_end = min(g.HW, p_begin + g.ppb_apply);
    if (p_begin + pp < p_end) h_next = *reinterpret_cast<const Half8*>(src + (p_begin + pp) * xs);
  }
}

inline bool make_gn_geom(int N, int64_t HW, int C, int G, GnGeom& g) {
  if (N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G != 0 || C % 8 != 0) return false;
  g.C = C; g.G = G; g.cg = C / G; g.OC = C / 8; g.HW = HW;
  if (g.OC > 1024) return false;
  for (int o = 0; o < g.OC; ++o)                  // an octet may span at most two groups
    if ((8 * o + 7) / g.cg - (8 * o) / g.cg > 1) return false;
  g.PP = g.OC >= 256 ? 1 : 256 / g.OC;
  if (G > g.OC * g.PP) return false;              // the block reduces with G threads
  // Blocks per IMAGE: about 512, a whole number of block-iterations each -- a function of the image
  // alone, never of the batch: the summation order, hence every bit of the result, is the same
  // whether an image runs alone or as one row of a batch (what makes batch sharding over GPUs
  // exact).  (Fixed rule: the oracle restates it, because it fixes the summation order.)
  const int64_t target = 512;
  int64_t ppb = (HW + target - 1) / target;
  ppb = ((ppb + g.PP - 1) / g.PP) * g.PP;
  g.ppb = (int)ppb;
  g.nchunk = (int)((HW + ppb - 1) / ppb);
  g.ppb_apply = g.ppb;
  g.nchunk_apply = g.nchunk;
  g.OCs = g.OC; g.PPa = g.PP; g.GS = G;
  g.C1 = C;
  return true;
}

// ------------------------------------------------------------------------------- LayerNorm
// One wave per row.  Lane l owns the 8-element chunks l, l + 64, l + 128, ... (at most 4).
// The row statistics follow the tiling-independent order of oracle/mixdq_oracle.c (round 5), which the
// GEMM epilogue of csrc/igemm_ln.hip reproduces from ONE record per row and column tile: per 16-column
// group (a pair of neighbouring lanes) the sum and the centred sum of squares; `per` consecutive groups
// form a unit (80 columns at C = 1280 / 640: the GEMM's column tile) whose groups are folded left to right
// with Chan's combination (through a wave-private LDS row, one lane per unit); the U <= 16 units are
// combined by a balanced tree (xor butterfly among lanes 0 .. U-1).
constexpr int kLnMaxChunks = 4;   // C <= 2048
constexpr int kLnMaxGroups = 64 * kLnMaxChunks / 2;

// balanced tree over the values of lanes 0 .. U-1 (U a power of two <= 16): pairs at distance 1, 2, 4, 8 by DPP
// (quad permutes, then the half-row and row mirrors: after two levels a quad holds one value, after three a
// half-row does, so a mirror reads the value the xor partner holds) -- the pairs, hence the bits, of the xor
// butterfly this replaces (__shfl_xor is ds_bpermute, the LDS crossbar: twelve of them were on the row's critical
// path); every lane returns the total.
template <int CTRL>
__device__ __forceinline__ float ln_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ln_tree(float t, int U) {
  if (U > 1) t = __fadd_rn(t, ln_dpp_f<0xB1>(t));    // quad_perm [1,0,3,2]
  if (U > 2) t = __fadd_rn(t, ln_dpp_f<0x4E>(t));    // quad_perm [2,3,0,1]
  if (U > 4) t = __fadd_rn(t, ln_dpp_f<0x141>(t));   // row_half_mirror
  if (U > 8) t = __fadd_rn(t, ln_dpp_f<0x140>(t));   // row_mirror
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t)));
}
// NQ: quantizers in use (0..3), WANT_H: the FP16 copy is written -- compile-time, so a launch with one
// consumer does not run the other two quantizers' arithmetic on its one-wave-per-SIMD critical path.
// ROWS: rows a wave carries through the chain together (all loaded up front).  The kernel is a
// latency chain per wave (load -> reduce -> reduce -> store); at batch 1 every wave has one row and
// the chain IS the kernel, from 8192 rows on two rows per wave keep the grid at one round of four
// waves per SIMD and put twice the bytes in flight per chain (batch 8: 13.0 us per launch at
// 2.4 TB/s before).  The arithmetic of a row does not depend on ROWS.  (Round 4: a streamed form -- a
// wave walks 4 or 8 consecutive rows and requests row r + 1 before it reduces row r -- was built and
// measured SLOWER: [8192, 1280] 11.4 vs 9.9 us, [32768, 640] 28.2 vs 21.3: fewer, longer waves put fewer
// bytes in flight than one round of short ones.  Removed.)
template <bool UNFUSED, int NQ, bool WANT_H, int ROWS>
__global__ __launch_bounds__(256) void ln_quant_kernel(
    const __half* __restrict__ x, const __half* __restrict__ gamma, const __half* __restrict__ beta,
    float eps, int64_t M, int C, const float* __restrict__ s_inv0, const float* __restrict__ zp0,
    int8_t* __restrict__ q0, const float* __restrict__ s_inv1, const float* __restrict__ zp1,
    int8_t* __restrict__ q1, const float* __restrict__ s_inv2, const float* __restrict__ zp2,
    int8_t* __restrict__ q2, __half* __restrict__ out_h) {
  MIXDQ_ARGS_NOW(x, gamma, beta, eps, M, C, s_inv0, zp0, q0, s_inv1, zp1, q1, s_inv2, zp2, q2, out_h);
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= M) return;
  const int nch = C / 8;
  Half8 h[ROWS][kLnMaxChunks], gmv[kLnMaxChunks], btv[kLnMaxChunks];
  // everything the rows need is requested up front: gamma / beta / the quantizer scalars would
  // otherwise add a second memory round trip after the reductions
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const __half* xr = x + (row0 + r < M ? row0 + r : row0) * C;    // a missing last row re-reads row0
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) h[r][i] = *reinterpret_cast<const Half8*>(xr + 8 * c);
    }
  }
#pragma unroll
  for (int i = 0; i < kLnMaxChunks; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      gmv[i] = *reinterpret_cast<const Half8*>(gamma + 8 * c);
      btv[i] = *reinterpret_cast<const Half8*>(beta + 8 * c);
    }
  }
  const float si0 = NQ > 0 ? *s_inv0 : 0.f, z0 = NQ > 0 ? *zp0 : 0.f;
  const float si1 = NQ > 1 ? *s_inv1 : 0.f, z1 = NQ > 1 ? *zp1 : 0.f;
  const float si2 = NQ > 2 ? *s_inv2 : 0.f, z2 = NQ > 2 ? *zp2 : 0.f;
  float mean[ROWS], rstd[ROWS];
  __shared__ float ln_sh[4][ROWS][kLnMaxGroups + 16];
  const int wv = threadIdx.x >> 6;
  const int G = C / 16;
  int U = 1;
  while (U < 16 && G % (2 * U) == 0) U *= 2;
  const int per = G / U;
  const float n_u = (float)(16 * per);
  const float inv_nu = 1.0f / n_u, inv_c = 1.0f / (float)C;     // the specification multiplies by these
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float mg[kLnMaxChunks], m2[kLnMaxChunks];
    float* sh = ln_sh[wv][r];            // [G] group values, then [16] unit means
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int c = lane + 64 * i;
      mg[i] = m2[i] = 0.f;
      if (c < nch) {        // (nch is even: both lanes of a pair take the branch)
        float s8 = half_at(h[r][i], 0);
#pragma unroll
        for (int j = 1; j < 8; ++j) s8 = __fadd_rn(s8, half_at(h[r][i], j));
        const float s1 = __fadd_rn(s8, ln_dpp_f<0xB1>(s8));          // the group's two halves
        mg[i] = __fmul_rn(s1, 0.0625f);
        float q8 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = __fsub_rn(half_at(h[r][i], j), mg[i]);
          q8 = __builtin_fmaf(d, d, q8);
        }
        m2[i] = __fadd_rn(q8, ln_dpp_f<0xB1>(q8));
        if ((lane & 1) == 0) sh[c >> 1] = s1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // wave-private row: no block barrier
    float s1u = 0.f, mu = 0.f;
    if (lane < U) {                                                  // one lane per unit: its groups, left to right
      s1u = sh[lane * per];
      for (int g = 1; g < per; ++g) s1u = __fadd_rn(s1u, sh[lane * per + g]);
      mu = __fmul_rn(s1u, inv_nu);
      sh[kLnMaxGroups + lane] = mu;
    }
    mean[r] = __fmul_rn(ln_tree(s1u, U), inv_c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int c = lane + 64 * i;
      if (c < nch && (lane & 1) == 0) {
        const float e = __fsub_rn(mg[i], sh[kLnMaxGroups + (c >> 1) / per]);
        sh[c >> 1] = __builtin_fmaf(__fmul_rn(e, 16.0f), e, m2[i]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float du = 0.f;
    if (lane < U) {
      float m2u = sh[lane * per];
      for (int g = 1; g < per; ++g) m2u = __fadd_rn(m2u, sh[lane * per + g]);
      const float e = __fsub_rn(mu, mean[r]);
      du = __builtin_fmaf(__fmul_rn(e, n_u), e, m2u);
    }
    rstd[r] = 1.0f / sqrtf(__fadd_rn(__fmul_rn(ln_tree(du, U), inv_c), eps));
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (row0 + r >= M) break;
#pragma unroll
    for (int i = 0; i < kLnMaxChunks; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const Half8 gm = gmv[i], bt = btv[i];
        Half8 oh;
        oh.w[0] = oh.w[1] = oh.w[2] = oh.w[3] = 0;
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float nrm = __fmul_rn(__fsub_rn(half_at(h[r][i], j), mean[r]), rstd[r]);
          y[j] = round_f16(__builtin_fmaf(nrm, half_at(gm, j), half_at(bt, j)));
          if constexpr (WANT_H) put_half(oh, j, y[j]);
        }
        const int64_t off = (row0 + r) * C + 8 * c;     // (quantize_pack8: clamp and packing in 3 instructions per pair)
        if constexpr (NQ > 0) *reinterpret_cast<uint2*>(q0 + off) = quantize_pack8<UNFUSED>(y, si0, z0);
        if constexpr (NQ > 1) *reinterpret_cast<uint2*>(q1 + off) = quantize_pack8<UNFUSED>(y, si1, z1);
        if constexpr (NQ > 2) *reinterpret_cast<uint2*>(q2 + off) = quantize_pack8<UNFUSED>(y, si2, z2);
        if constexpr (WANT_H) *reinterpret_cast<Half8*>(out_h + off) = oh;
      }
    }
  }
}

// ------------------------------------------------------------------------------- GEGLU
// h [M, 2D] (ff.net.0.proj output): y = fp16(fp16(h[:, :D]) * fp16(gelu(h[:, D:]))) -> quantize.
template <bool UNFUSED>
__global__ __launch_bounds__(256) void geglu_quant_kernel(
    const __half* __restrict__ h, int64_t M, int D, const float* __restrict__ s_inv_p,
    const float* __restrict__ zp_p, int8_t* __restrict__ out_q, __half* __restrict__ out_h) {
  const int dch = D / 8;
  const int64_t total = M * dch;
  const bool want_q = out_q != nullptr;
  const float s_inv = want_q ? *s_inv_p : 0.f, zp = want_q ? *zp_p : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / dch;
    const int c = (int)(i - m * dch);
    const Half8 xv = *reinterpret_cast<const Half8*>(h + m * 2 * D + 8 * c);
    const Half8 gv = *reinterpret_cast<const Half8*>(h + m * 2 * D + D + 8 * c);
    Half8 oh;
    oh.w[0] = oh.w[1] = oh.w[2] = oh.w[3] = 0;
    float y8[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {       // two gate values at a time (packed FP32: geluf2)
      const v2f g2 = geluf2(v2f{half_at(gv, j), half_at(gv, j + 1)});
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float ge = round_f16(g2[e]);
        const float y = round_f16(__fmul_rn(half_at(xv, j + e), ge));
        put_half(oh, j + e, y);
        y8[j + e] = y;
      }
    }
    if (want_q) *reinterpret_cast<uint2*>(out_q + m * D + 8 * c) = quantize_pack8<UNFUSED>(y8, s_inv, zp);
    if (out_h) *reinterpret_cast<Half8*>(out_h + m * D + 8 * c) = oh;
  }
}

// The same pass with GELU by table (round 6; the stand-alone GEGLU launch of the module-swap path and of the layers
// the fused graph does not fuse): f16(gelu(g)) for an FP16 gate is a function of 16 bits -- igemm's GEMM + GEGLU
// epilogue has looked it up since round 3 (csrc/igemm_kernel.h: |g| < 16, both signs, 76 KB, built by the scalar
// specification); this translation unit keeps its own copy of that table, built the same way, and 1024-thread blocks
// (two per CU: 37 registers) copy it into LDS by LDS-DMA at entry and then stride over the tensor.  A gate outside the table, NaN or
// +-inf takes the arithmetic (mixdq_geluf), decided per wave.  Bit-identical by construction.
constexpr int kGeluMagFn = 0x4c00;                        // |g| < 16.0, as kGeluTabMag of csrc/igemm_kernel.h
constexpr int kGeluTabBytesFn = 2 * kGeluMagFn * 2;
__device__ uint16_t g_gelu_tab_fn[2 * kGeluMagFn];

__global__ __launch_bounds__(256) void gelu_table_fn_init_kernel() {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * kGeluMagFn) return;
  __half_raw r;
  r.x = (unsigned short)((i >= kGeluMagFn ? 0x8000 : 0) | (i % kGeluMagFn));
  g_gelu_tab_fn[i] = __half_as_ushort(f32_to_f16_rn(mixdq_geluf(__half2float(__half(r)))));
}

inline int ensure_gelu_table_fn(hipStream_t stream) {      // (the logic of ensure_silu_table)
  static bool done[64] = {};
  static unsigned long long in_capture[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
  if (done[dev]) return MIXDQ_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  unsigned long long cap_id = 0;
  if (hipStreamGetCaptureInfo(stream, &cap, &cap_id) != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap != hipStreamCaptureStatusNone && cap_id != 0 && in_capture[dev] == cap_id) return MIXDQ_OK;
  gelu_table_fn_init_kernel<<<(2 * kGeluMagFn + 255) / 256, 256, 0, stream>>>();
  if (hipGetLastError() != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap == hipStreamCaptureStatusNone) {
    if (hipStreamSynchronize(stream) != hipSuccess) return MIXDQ_ERR_LAUNCH;
    done[dev] = true;
  } else {
    in_capture[dev] = cap_id;
  }
  return MIXDQ_OK;
}

template <bool UNFUSED>
__global__ __launch_bounds__(1024) void geglu_quant_tab_kernel(
    const __half* __restrict__ h, int64_t M, int D, const float* __restrict__ s_inv_p,
    const float* __restrict__ zp_p, int8_t* __restrict__ out_q, __half* __restrict__ out_h) {
  extern __shared__ __attribute__((aligned(16))) char gg_dyn[];
  {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int q = wave; q < kGeluTabBytesFn / 1024; q += 16)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g_gelu_tab_fn) + q * 1024 +
                                                          ((int)threadIdx.x & 63) * 16),
          (__attribute__((address_space(3))) void*)(gg_dyn + q * 1024), 16, 0, 0);
  }
  const int dch = D / 8;
  const int64_t total = M * dch;
  const bool want_q = out_q != nullptr;
  using cf32 = const __attribute__((address_space(4))) float;
  const float s_inv = want_q ? *(cf32*)s_inv_p : 0.f, zp = want_q ? *(cf32*)zp_p : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint16_t* tab = reinterpret_cast<const uint16_t*>(gg_dyn);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / dch;
    const int c = (int)(i - m * dch);
    const Half8 xv = *reinterpret_cast<const Half8*>(h + m * 2 * D + 8 * c);
    const Half8 gv = *reinterpret_cast<const Half8*>(h + m * 2 * D + D + 8 * c);
    uint32_t ge[8];
    bool far = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t gb = (j & 1) ? (gv.w[j >> 1] >> 16) : (gv.w[j >> 1] & 0xffffu);
      const uint32_t mag = gb & 0x7fffu;
      const bool in = mag < (uint32_t)kGeluMagFn;
      far |= !in;
      ge[j] = tab[in ? (gb >> 15) * kGeluMagFn + mag : 0u];            // f16(gelu(g)), from the table
    }
    if (__builtin_amdgcn_ballot_w64(far) != 0) {                       // wave-uniform, rare: beyond the table
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t gb = (j & 1) ? (gv.w[j >> 1] >> 16) : (gv.w[j >> 1] & 0xffffu);
        if ((gb & 0x7fffu) >= (uint32_t)kGeluMagFn)
          ge[j] = __half_as_ushort(f32_to_f16_rn(mixdq_geluf(half_at(gv, j))));   // the specification itself
      }
    }
    Half8 oh;
    oh.w[0] = oh.w[1] = oh.w[2] = oh.w[3] = 0;
    float y8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __half_raw r;
      r.x = (unsigned short)ge[j];
      const float y = round_f16(__fmul_rn(half_at(xv, j), __half2float(__half(r))));
      put_half(oh, j, y);
      y8[j] = y;
    }
    if (want_q) *reinterpret_cast<uint2*>(out_q + m * D + 8 * c) = quantize_pack8<UNFUSED>(y8, s_inv, zp);
    if (out_h) *reinterpret_cast<Half8*>(out_h + m * D + 8 * c) = oh;
  }
}

}  // namespace
}  // namespace mixdq

using namespace mixdq;

extern "C" size_t mixdq_groupnorm_workspace_bytes(int N, int64_t HW, int C, int G) {
  GnGeom g;
  if (!make_gn_geom(N, HW, C, G, g)) return 0;
  return ((size_t)N * g.nchunk * G + (size_t)N * G) * sizeof(float2);
}

extern "C" int mixdq_groupnorm_silu_quantize2(const void* x_nhwc, int C1, const void* x2_nhwc,
                                              const void* gamma, const void* beta, float eps,
                                              int apply_silu, const float* scale_inv,
                                              const float* zero_point, int8_t* out_q_or_null,
                                              void* out_f16_or_null, void* workspace, int N,
                                              int64_t HW, int C, int G, int flags,
                                              mixdq_stream_t stream_);
extern "C" int mixdq_groupnorm_silu_quantize3(const void* x_nhwc, int C1, const void* x2_nhwc,
                                              const void* gamma, const void* beta, float eps,
                                              int apply_silu, const float* scale_inv,
                                              const float* zero_point, int8_t* out_q_or_null,
                                              void* out_f16_or_null,
                                              const float* const* raw_scale_inv,
                                              const float* const* raw_zero_point,
                                              int8_t* const* raw_q, void* workspace, int N,
                                              int64_t HW, int C, int G, int flags,
                                              mixdq_stream_t stream_);

extern "C" int mixdq_groupnorm_silu_quantize(const void* x_nhwc, const void* gamma,
                                             const void* beta, float eps, int apply_silu,
                                             const float* scale_inv, const float* zero_point,
                                             int8_t* out_q_or_null, void* out_f16_or_null,
                                             void* workspace, int N, int64_t HW, int C, int G,
                                             int flags, mixdq_stream_t stream_) {
  return mixdq_groupnorm_silu_quantize2(x_nhwc, C, nullptr, gamma, beta, eps, apply_silu, scale_inv,
                                        zero_point, out_q_or_null, out_f16_or_null, workspace, N, HW,
                                        C, G, flags, stream_);
}

extern "C" int mixdq_groupnorm_silu_quantize2(const void* x_nhwc, int C1, const void* x2_nhwc,
                                              const void* gamma, const void* beta, float eps,
                                              int apply_silu, const float* scale_inv,
                                              const float* zero_point, int8_t* out_q_or_null,
                                              void* out_f16_or_null, void* workspace, int N,
                                              int64_t HW, int C, int G, int flags,
                                              mixdq_stream_t stream_) {
  return mixdq_groupnorm_silu_quantize3(x_nhwc, C1, x2_nhwc, gamma, beta, eps, apply_silu, scale_inv,
                                        zero_point, out_q_or_null, out_f16_or_null, nullptr, nullptr,
                                        nullptr, workspace, N, HW, C, G, flags, stream_);
}

extern "C" int mixdq_groupnorm_silu_quantize3(const void* x_nhwc, int C1, const void* x2_nhwc,
                                              const void* gamma, const void* beta, float eps,
                                              int apply_silu, const float* scale_inv,
                                              const float* zero_point, int8_t* out_q_or_null,
                                              void* out_f16_or_null,
                                              const float* const* raw_scale_inv,
                                              const float* const* raw_zero_point,
                                              int8_t* const* raw_q, void* workspace, int N,
                                              int64_t HW, int C, int G, int flags,
                                              mixdq_stream_t stream_) {
  GnRaw raw = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  if (raw_q != nullptr) {
    for (int i = 0; i < 2; ++i) {
      if (raw_q[i] == nullptr) continue;
      if (i == 1 && x2_nhwc == nullptr) return MIXDQ_ERR_INVALID_ARG;   // no second source
      if (!raw_scale_inv || !raw_zero_point || !raw_scale_inv[i] || !raw_zero_point[i])
        return MIXDQ_ERR_INVALID_ARG;
      if ((uintptr_t)raw_q[i] % 8) return MIXDQ_ERR_ALIGNMENT;
      raw.s_inv[i] = raw_scale_inv[i]; raw.zp[i] = raw_zero_point[i]; raw.q[i] = raw_q[i];
    }
  }
  GnGeom g;
  if (!make_gn_geom(N, HW, C, G, g)) return MIXDQ_ERR_SHAPE;
  if (C1 <= 0 || C1 > C || C1 % 8 != 0) return MIXDQ_ERR_SHAPE;
  if ((C1 < C) != (x2_nhwc != nullptr) || ((uintptr_t)x2_nhwc % 16)) return MIXDQ_ERR_INVALID_ARG;
  g.C1 = C1;
  if (!x_nhwc || !gamma || !beta || !workspace || (!out_q_or_null && !out_f16_or_null))
    return MIXDQ_ERR_INVALID_ARG;
  if (out_q_or_null && (!scale_inv || !zero_point)) return MIXDQ_ERR_INVALID_ARG;
  if (((uintptr_t)x_nhwc | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_f16_or_null) % 16 ||
      (uintptr_t)out_q_or_null % 8)
    return MIXDQ_ERR_ALIGNMENT;
  hipStream_t stream = (hipStream_t)stream_;
  float2* partial = (float2*)workspace;
  const int threads = g.OC * g.PP;
  static const int stats_unroll = [] { const char* e = getenv("MIXDQ_GN_STATS_UNROLL"); return e ? atoi(e) : 0; }();
  const int su = stats_unroll == 1 || stats_unroll == 2 || stats_unroll == 4 ? stats_unroll : (g.ppb / g.PP <= 2 ? 2 : 4);
  (su == 1 ? gn_stats_kernel<1> : su == 2 ? gn_stats_kernel<2> : gn_stats_kernel<4>)
      <<<dim3(g.nchunk, N), threads, threads * 4 * sizeof(float), stream>>>(
          (const __half*)x_nhwc, (const __half*)x2_nhwc, partial, g);
  // MIXDQ_GN_SLICED=1 (measured, off): with more than 64 partials per group the apply pass is cut into channel
  // slices (blockIdx.z) of whole groups, at most 8 groups each, so that a block needs -- and reduces in its
  // prologue -- only its slice's statistics (4 x 342 x 8 B at C = 640) and the finalize launch disappears (46 per
  // batch-1 step).  Bit-exact, and slower: slices are 160-320 contiguous bytes per pixel (80-byte INT8 rows), the
  // prologue's strided partial reads are one cache line per lane: batch 1 11.08 -> 11.24 ms, batch 8 47.5 -> 48.9
  // (profiles/r05_gn_finalize_ab.txt).
  static const bool sliced_on = [] { const char* e = getenv("MIXDQ_GN_SLICED"); return e && e[0] == '1'; }();
  int nslice = 1;
  if (g.nchunk > 64 && sliced_on)
    for (int s = 8; s >= 2 && nslice == 1; s >>= 1)
      if (g.OC % s == 0 && G % s == 0 && G / s <= 8 && g.OC / s >= 10) nslice = s;
  float2* stats = nullptr;          // null: the apply blocks finalize themselves
  if (g.nchunk > 64 && nslice == 1) {
    stats = partial + (size_t)N * g.nchunk * G;
    gn_finalize_kernel<<<dim3(G, N), 64, 0, stream>>>(partial, stats, g, eps);
  }
  // The apply pass is elementwise (any split gives the same bits).  With SiLU it is a VALU chain per lane
  // (exp and a correctly rounded division per element); at small batch -- fewer than 1024 statistics blocks,
  // every block's chain on the launch's critical path -- half-length blocks end sooner: (1, 128 x 128, 960)
  // 42.5 -> 38.3 us for the three launches, (1, 64 x 64, 1920) 26.6 -> 25.5, (1, 32 x 32, 1280) 13.6 -> 13.0.
  // Shorter still, or from batch 2 on, or without SiLU: equal or slower.  From 2048 blocks on, FEWER and longer
  // blocks are slower too ((8, 128 x 128, 320): 75 us at 512 per image, 80 at 256, 99 at 64), and requesting
  // the next pixel one iteration ahead changes nothing (tools/gpu_s12.sh, tools/gpu_s13.sh).
  if (apply_silu && (int64_t)N * g.nchunk < 1024 && g.ppb >= 2 * g.PP) {
    g.ppb_apply = (g.ppb / 2 / g.PP) * g.PP;
    g.nchunk_apply = (int)((HW + g.ppb_apply - 1) / g.ppb_apply);
  }
  int threads_apply = threads;
  if (nslice > 1) {   // the same number of pixels per THREAD as the unsliced pass would take
    const int iters = g.ppb_apply / g.PP;
    g.OCs = g.OC / nslice; g.GS = G / nslice;
    g.PPa = (256 + g.OCs - 1) / g.OCs;          // >= 256 threads: four complete waves for the prologue
    g.ppb_apply = iters * g.PPa;
    g.nchunk_apply = (int)((HW + g.ppb_apply - 1) / g.ppb_apply);
    threads_apply = g.OCs * g.PPa;
  }
  const bool unfused = flags & MIXDQ_FLAG_UNFUSED;
  // SiLU by table (see gn_apply_kernel): where the launch is large enough to pay for 75 KB of table per block -- from
  // MIXDQ_GN_SILU_TAB_MIN elements on (default 2 Mi) -- statistics from the finalize launch, no channel slices.
  // Measured (tools/bench_norms.py, profiles/r06_gn_silu_table_ab.txt; us for the three launches, arithmetic -> table):
  // (8, 128 x 128, 320) 76.2 -> 57.2, (8, 64 x 64, 640) 46.8 -> 37.0, (8, 32 x 32, 1280) 32.7 -> 26.6; one image:
  // (1, 128 x 128, 960) 40.0 -> 30.4, (1, 64 x 64, 1920) 26.2 -> 20.2, (1, 128 x 128, 320) 21.6 -> 18.3, (1, 64 x 64, 640)
  // 15.9 -> 14.9 -- and (1, 32 x 32, 1280), 1.3 Mi elements, 12.8 -> 13.3: the threshold.
  // MIXDQ_GN_SILU_TAB=0 never / 1 wherever possible (tests).
  static const int tab_mode = [] { const char* e = getenv("MIXDQ_GN_SILU_TAB"); return e ? atoi(e) : -1; }();
  static const int64_t tab_min = [] { const char* e = getenv("MIXDQ_GN_SILU_TAB_MIN"); return e ? atoll(e) : (int64_t)2 << 20; }();
  const bool use_tab = apply_silu && stats != nullptr && nslice == 1 && tab_mode != 0 &&
                       (tab_mode == 1 || (int64_t)N * HW * C >= tab_min);
  if (use_tab) {
    if (const int st = ensure_silu_table(stream)) return st;
    // ~512 threads per block, two blocks per CU (75 KB of LDS each), every block walking several chunks of its image;
    // the same number of pixels per THREAD and chunk as the plain pass would take (the pass is elementwise: any split
    // gives the same bits)
    const int iters = g.ppb_apply / g.PP;
    g.PPa = g.OC >= 512 ? 1 : 512 / g.OC;
    g.ppb_apply = iters * g.PPa;
    g.nchunk_apply = (int)((HW + g.ppb_apply - 1) / g.ppb_apply);
    const int threads_tab = g.OC * g.PPa;
    int gx = (2 * kNumCU + N - 1) / N;
    if (gx > g.nchunk_apply) gx = g.nchunk_apply;
    if (gx < 1) gx = 1;
    static bool seen_t[2][64] = {};
#define GN_APPLY_TAB(U)                                                                                         \
    do {                                                                                                        \
      if (const int st = lds_opt_in(reinterpret_cast<const void*>(&gn_apply_kernel<true, U, false, true>),       \
                                    kSiluTabBytes, seen_t[U ? 1 : 0])) return st;                                \
      gn_apply_kernel<true, U, false, true><<<dim3(gx, N, 1), threads_tab, kSiluTabBytes, stream>>>(            \
          (const __half*)x_nhwc, (const __half*)x2_nhwc, partial, stats, eps, (const __half*)gamma,             \
          (const __half*)beta, scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null, g, raw);         \
    } while (0)
    if (unfused) GN_APPLY_TAB(true); else GN_APPLY_TAB(false);
#undef GN_APPLY_TAB
    return launch_status();
  }
  const dim3 grid(g.nchunk_apply, N, nslice);
#define GN_APPLY(S, U)                                                                          \
  (stats ? gn_apply_kernel<S, U, false> : gn_apply_kernel<S, U, true>)<<<grid, threads_apply, 0, stream>>>( \
      (const __half*)x_nhwc, (const __half*)x2_nhwc, partial, stats, eps, (const __half*)gamma, \
      (const __half*)beta,                                                                      \
      scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null, g, raw)
  if (apply_silu) { if (unfused) GN_APPLY(true, true); else GN_APPLY(true, false); }
  else            { if (unfused) GN_APPLY(false, true); else GN_APPLY(false, false); }
#undef GN_APPLY
  return launch_status();
}

// The FP16 -> FP16 SiLU table of the apply pass's TAB variant (built on first use; this call builds it eagerly --
// mixdq_amd does so before it captures a graph -- and copies it out for the test that holds every entry to the
// scalar specification): [kSiluPos] entries for y = +bits, then [kSiluNeg] for y = -bits.
extern "C" int mixdq_silu_table(uint16_t* out_device_or_null, int* n_pos, int* n_neg, mixdq_stream_t stream) {
  if (n_pos) *n_pos = kSiluPos;
  if (n_neg) *n_neg = kSiluNeg;
  if (const int st = ensure_silu_table((hipStream_t)stream)) return st;
  if (!out_device_or_null) return MIXDQ_OK;
  return hipMemcpyFromSymbolAsync(out_device_or_null, HIP_SYMBOL(g_silu_tab), 2 * (kSiluPos + kSiluNeg), 0,
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess
             ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}

extern "C" int mixdq_layernorm_quantize(const void* x, const void* gamma, const void* beta,
                                        float eps, int64_t M, int C, int n_out,
                                        const float* const* scale_inv,
                                        const float* const* zero_point, int8_t* const* out_q,
                                        void* out_f16_or_null, int flags, mixdq_stream_t stream_) {
  if (M < 0 || C <= 0 || n_out < 0 || n_out > 3) return MIXDQ_ERR_INVALID_ARG;
  if (C % 16 != 0 || C / 8 > 64 * kLnMaxChunks) return MIXDQ_ERR_SHAPE;   // (16-column groups: the statistics' order)
  if (M == 0) return MIXDQ_OK;
  if (!x || !gamma || !beta || (n_out == 0 && !out_f16_or_null)) return MIXDQ_ERR_INVALID_ARG;
  const float* si[3] = {nullptr, nullptr, nullptr};
  const float* zp[3] = {nullptr, nullptr, nullptr};
  int8_t* q[3] = {nullptr, nullptr, nullptr};
  for (int i = 0; i < n_out; ++i) {
    if (!scale_inv || !zero_point || !out_q || !scale_inv[i] || !zero_point[i] || !out_q[i])
      return MIXDQ_ERR_INVALID_ARG;
    si[i] = scale_inv[i]; zp[i] = zero_point[i]; q[i] = out_q[i];
    if ((uintptr_t)q[i] % 8) return MIXDQ_ERR_ALIGNMENT;
  }
  if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out_f16_or_null) % 16)
    return MIXDQ_ERR_ALIGNMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const bool unfused = flags & MIXDQ_FLAG_UNFUSED, want_h = out_f16_or_null != nullptr;
  static const int rows_forced = [] { const char* e = getenv("MIXDQ_LN_ROWS"); return e ? atoi(e) : 0; }();
  const int rows = rows_forced == 1 || rows_forced == 2 ? rows_forced   // (tuning runs)
                                                        : (M >= 8192 ? 2 : 1);   // rows per wave (see the kernel)
  const int grid = (int)((M + 4 * rows - 1) / (4 * rows));
#define LN_LAUNCH(U, NQ, H)                                                                          \
  do {                                                                                               \
    if (rows == 2)                                                                                   \
      ln_quant_kernel<U, NQ, H, 2><<<grid, 256, 0, stream>>>(                                        \
          (const __half*)x, (const __half*)gamma, (const __half*)beta, eps, M, C, si[0], zp[0], q[0], \
          si[1], zp[1], q[1], si[2], zp[2], q[2], (__half*)out_f16_or_null);                         \
    else                                                                                             \
      ln_quant_kernel<U, NQ, H, 1><<<grid, 256, 0, stream>>>(                                        \
          (const __half*)x, (const __half*)gamma, (const __half*)beta, eps, M, C, si[0], zp[0], q[0], \
          si[1], zp[1], q[1], si[2], zp[2], q[2], (__half*)out_f16_or_null);                         \
  } while (0)
#define LN_BY_H(U, NQ) do { if (want_h) LN_LAUNCH(U, NQ, true); else LN_LAUNCH(U, NQ, false); } while (0)
#define LN_BY_NQ(U)                                                                \
  do {                                                                             \
    switch (n_out) {                                                               \
      case 0: LN_LAUNCH(U, 0, true); break;   /* n_out == 0 implies the FP16 copy */ \
      case 1: LN_BY_H(U, 1); break;                                                \
      case 2: LN_BY_H(U, 2); break;                                                \
      default: LN_BY_H(U, 3); break;                                               \
    }                                                                              \
  } while (0)
  if (unfused) LN_BY_NQ(true); else LN_BY_NQ(false);
#undef LN_BY_NQ
#undef LN_BY_H
#undef LN_LAUNCH
  return launch_status();
}

extern "C" int mixdq_geglu_quantize(const void* h, int64_t M, int D, const float* scale_inv,
                                    const float* zero_point, int8_t* out_q_or_null,
                                    void* out_f16_or_null, int flags, mixdq_stream_t stream_) {
  if (M < 0 || D <= 0) return MIXDQ_ERR_INVALID_ARG;
  if (D % 8 != 0) return MIXDQ_ERR_SHAPE;
  if (M == 0) return MIXDQ_OK;
  if (!h || (!out_q_or_null && !out_f16_or_null)) return MIXDQ_ERR_INVALID_ARG;
  if (out_q_or_null && (!scale_inv || !zero_point)) return MIXDQ_ERR_INVALID_ARG;
  if (((uintptr_t)h | (uintptr_t)out_f16_or_null) % 16 || (uintptr_t)out_q_or_null % 8)
    return MIXDQ_ERR_ALIGNMENT;
  int64_t blocks = (M * (D / 8) + 255) / 256;
  if (blocks > kNumCU * 8) blocks = kNumCU * 8;
  hipStream_t stream = (hipStream_t)stream_;
  // GELU by table from 2 Mi outputs on (MIXDQ_GEGLU_TAB=0 never / 1 always: tests and A/B runs): (1024, 5120), the
  // batch-1 ff layer of the module-swap path, is 5 Mi
  static const int tab_mode = [] { const char* e = getenv("MIXDQ_GEGLU_TAB"); return e ? atoi(e) : -1; }();
  if (tab_mode != 0 && (tab_mode == 1 || M * (int64_t)D >= ((int64_t)2 << 20))) {
    if (const int st = ensure_gelu_table_fn(stream)) return st;
    static bool seen_g[2][64] = {};
    const bool unf = flags & MIXDQ_FLAG_UNFUSED;
    const void* kern = unf ? reinterpret_cast<const void*>(&geglu_quant_tab_kernel<true>)
                           : reinterpret_cast<const void*>(&geglu_quant_tab_kernel<false>);
    if (const int st = lds_opt_in(kern, kGeluTabBytesFn, seen_g[unf ? 1 : 0])) return st;
    int64_t tb = (M * (D / 8) + 1023) / 1024;
    if (tb > 2 * kNumCU) tb = 2 * kNumCU;
    if (unf)
      geglu_quant_tab_kernel<true><<<(int)tb, 1024, kGeluTabBytesFn, stream>>>(
          (const __half*)h, M, D, scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null);
    else
      geglu_quant_tab_kernel<false><<<(int)tb, 1024, kGeluTabBytesFn, stream>>>(
          (const __half*)h, M, D, scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null);
    return launch_status();
  }
  if (flags & MIXDQ_FLAG_UNFUSED)
    geglu_quant_kernel<true><<<(int)blocks, 256, 0, stream>>>(
        (const __half*)h, M, D, scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null);
  else
    geglu_quant_kernel<false><<<(int)blocks, 256, 0, stream>>>(
        (const __half*)h, M, D, scale_inv, zero_point, out_q_or_null, (__half*)out_f16_or_null);
  return launch_status();
}
