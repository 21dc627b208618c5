// a3 for the UNet's dominant convolution shape -- 3x3, stride 1, pad 1, C % 64 == 0 -- with the INPUT
// HALO resident in LDS.  Replaces the same reference files as csrc/igemm.hip's CONV path
// (qconv2d.cc:27-206, cutlassConv2d_*.cu, conv_act_zero_point_propagate.cu) and produces the same
// bits (integer accumulation is exact and order-free; the epilogue arithmetic is igemm.hip's).
//
// Why a second kernel: the implicit-GEMM path gathers the activation operand once per TAP -- nine
// shifted copies of (almost) the same pixels per channel chunk -- and a CU's L2 -> LDS stream
// (~40-50 GB/s, what every tile of that family is bound by) carries 9 x (BM + BN) x 128 bytes per
// chunk.  Here a workgroup owns a TH x TW patch of output pixels: per 128-byte channel chunk it
// stages the (TH + 2) x (TW + 2) input halo ONCE (22.5 KB for 8 x 16 pixels) and streams only the
// weights, one filter row (3 taps x BN channels) per step; the nine taps read the same halo at
// lane-uniform pixel shifts.  Bytes per chunk: 22.5 + 9 x 10 = 113 KB against 9 x 26 = 234 KB
// (128 x 80 tile); with 16 x 16 pixels per workgroup (large batches) 65 KB per 128 x 80 outputs.
// K order becomes (channel chunk, r, s) instead of (r, s, channel): exact either way.
//
//   LDS: three weight stages (one filter row each; ring: row r always lives in stage r) + two halo
//        buffers (chunk parity), images [rows][128 B] with the 16-byte chunk XOR-swizzled by
//        (row >> 1) & 7 on the DMA source address and on the fragment read (both-sides rule).
//   Pipeline: LDS-DMA (global_load_lds_dwordx4), two weight stages and the next halo in flight, one
//        counted vmcnt + one s_barrier per filter row; past the end the DMAs read a zero page so the
//        counts stay uniform.
//   Waves: 4 (pixels) x 2 k-split groups, v_mfma_i32_16x16x64_i8, weights as the A operand: a lane
//        holds one pixel and 4 consecutive channels per register quad, as in igemm.hip.
#include <type_traits>

#include "iconv.h"

namespace mixdq {
namespace {

__device__ uint4 g_zero_page;   // device globals are zero-initialised

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ uint32_t add_f16x2_(uint32_t a, uint32_t b) {
  const v2h ah = *reinterpret_cast<const v2h*>(&a), bh = *reinterpret_cast<const v2h*>(&b);
  v2f r = __builtin_convertvector(ah, v2f) + __builtin_convertvector(bh, v2f);
  asm("" : "+v"(r));
  const v2h h = __builtin_convertvector(r, v2h);
  return *reinterpret_cast<const uint32_t*>(&h);
}

// CK: bytes of a channel chunk (a row of the LDS images): 128 with two k-split wave groups (each takes
// one 64-byte MFMA k-step of a chunk; 4 waves x BM / 4 pixels), or 64 with eight waves x BM / 8 pixels.
// WNG: wave groups over the BN channels (1: every wave computes all BN channels of its pixels; 2: a wave
// computes BN / 2 channels of twice the pixels -- 9 fragment reads per 20 MFMAs instead of 7 per 10).
template <int TH, int TW, int BN, int CK, int WNG = 1>
struct HaloGeom {
  static constexpr int BM = TH * TW;                                  // output pixels per workgroup
  static constexpr int HWP = TW + 2, HP = (TH + 2) * HWP;             // halo row length, halo pixels
  static constexpr int KSPLIT = CK / 64;                              // k-split wave groups
  static constexpr int NWAVES = 8, NTHREADS = 512, WPX = NWAVES / KSPLIT / WNG;   // waves over the pixels
  static constexpr int PPP = 1024 / CK, LPP = CK / 16;                // pixels per 1-KiB DMA piece; lanes per pixel
  static constexpr int H_NI = ((HP + PPP - 1) / PPP + NWAVES - 1) / NWAVES;   // halo pieces per wave
  static constexpr int HALO_BYTES = H_NI * NWAVES * 1024;
  static constexpr int W_TAP = BN * CK, W_STAGE = 3 * W_TAP;          // one tap tile; one filter row
  static constexpr int W_PIECES = W_STAGE / 1024;
  static constexpr int W_LO = W_PIECES / NWAVES, W_REM = W_PIECES % NWAVES, W_NI = W_LO + (W_REM ? 1 : 0);
  static constexpr int HALO_OFF = 3 * W_STAGE;
  static constexpr int MAIN_BYTES = HALO_OFF + 2 * HALO_BYTES;
  static constexpr int CS_STRIDE = BN * 2 + 16;                       // epilogue tile row stride
  static constexpr int WTN = BN / WNG;                                // channels per wave
  static constexpr int WTM = BM / WPX, TM = WTM / 16, TN = WTN / 16;  // wave tile: WTM pixels x WTN channels
  static constexpr int PART_BYTES = (KSPLIT - 1) * WPX * TM * TN * 4 * 64 * 4;   // k-split partials
  static constexpr int SMEM = MAIN_BYTES + 9 * BN * 4 + BN * 6;       // + 9 border-class rows, scale, bias
  static_assert(CK == 64 || CK == 128, "one or two 64-byte MFMA k-steps per chunk");
  static_assert(W_TAP % 1024 == 0 && WTN % 16 == 0 && WTM % 16 == 0, "whole DMA pieces / MFMA tiles");
  static_assert(WNG == 1 || KSPLIT == 1, "channel wave groups only without the k-split");
  static_assert(BM * CS_STRIDE + PART_BYTES <= MAIN_BYTES, "epilogue staging overlays the main buffers");
  static_assert(SMEM <= 160 * 1024, "LDS is 160 KiB per CU");
  static_assert(H_NI + 2 * W_NI <= 63, "vmcnt is a 6-bit counter");
  // 16-byte chunk swizzle of an image row (256-byte LDS bank rows hold 2 or 4 image rows).  A
  // ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
  // and the same + 32 -- i.e. with one pixel per lane & 15 and the k-chunk on lane >> 4, rows
  // b .. b+3 and b+12 .. b+15 with chunk c next to rows b+4 .. b+11 with chunk c ^ 1.  The taps
  // shift b by any amount, so the XOR term must separate those 16 lanes for EVERY b: the terms below
  // do (exhaustive check over b, both k-split groups); igemm.hip's (row >> 2) & 3 / (row >> 1) & 7,
  // which are conflict-free for its 32-row lane mapping and for b % 16 == 0, cost 1.9x / 1.7x the LDS
  // cycles here (SQ_LDS_BANK_CONFLICT 0.49 of SQ_LDS_IDX_ACTIVE on the 16 x 16 / 64-byte tile,
  // profiles/r04_pmc_summary.json).  Patches 8 pixels wide (two image rows per fragment) keep the old
  // term: no XOR of row bits does better than 1.8x there.
  __device__ static int swz(int row) {
    if (CK == 64) return ((row >> 2) & 1) << 1;
    return TW == 16 ? ((row >> 1) & 3) << 1 : (row >> 1) & 7;
  }
};

template <int TH, int TW, int BN, int CK, int WNG = 1>
__global__ __launch_bounds__(512, 2) void conv3x3_halo_kernel(const HaloConvArgs p) {
  using G = HaloGeom<TH, TW, BN, CK, WNG>;
  constexpr int BM = G::BM, HWP = G::HWP, HP = G::HP, TM = G::TM, TN = G::TN;
  constexpr int CS_STRIDE = G::CS_STRIDE;
  MIXDQ_ARGS_NOW(p.X, p.Wt, p.scale, p.bias, p.table, p.zp, p.D, p.res, p.res_div, p.NI, p.H, p.W, p.C,
                 p.K, p.unfused, p.ups);
  // the activation zero point: a SCALAR load at entry (constant address space: it does not change while the
  // kernel runs) -- read in front of the accumulator pass it was a dependent trip to memory in every launch
  const float zpv = *(const __attribute__((address_space(4))) float*)p.zp;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = WNG == 1 ? wid / G::WPX : 0;         // k-split group (0 when CK == 64)
  const int wn = WNG == 1 ? 0 : wid / G::WPX;         // channel group
  const int wm = wid % G::WPX;                        // pixel group
  const int lrow = lane & 15, lkq = lane >> 4;

  // ---- XCD-aware tile map (as igemm.hip): every XCD gets a contiguous run of the tile sequence,
  //      ordered in super-rows of 8 pixel tiles (pixel tile fastest, then channel tile)
  const int tiles_x = p.W / TW, tiles_y = p.H / TH, tiles_n = (p.K + BN - 1) / BN;
  const int tiles_m = p.NI * tiles_y * tiles_x;
  const int nwg = tiles_m * tiles_n;
  const int bid = blockIdx.x;
  const int xcd = bid % kNumXCD, q8 = nwg / kNumXCD, r8 = nwg % kNumXCD;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kNumXCD;
  constexpr int GM = 8;
  const int per_group = GM * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, tiles_m - first_m);
  const int rem = wg - group * per_group;
  const int tile_n = rem / gsz, tile_m = first_m + (rem - tile_n * gsz);
  const int img = tile_m / (tiles_y * tiles_x);
  const int trem = tile_m - img * (tiles_y * tiles_x);
  const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
  const int n0 = tile_n * BN;
  const int C = p.C;
  const int nch = (C + CK - 1) / CK;                 // channel chunks (the last may be half empty)
  const char* zero = reinterpret_cast<const char*>(&g_zero_page);
  // ups: the stored image is half the size; halo pixel (iy, ix) reads source pixel (iy / 2, ix / 2)
  const int sh = p.ups ? 1 : 0, SW = p.W >> sh;
  const int8_t* ximg = p.X + (int64_t)img * (p.H >> sh) * SW * C;

  // ---- epilogue vectors, requested now and parked in registers across the main loop (igemm.hip):
  //      the NINE border-class rows a 3x3 / pad-1 conv can meet (row class x column class: full
  //      window, first row / column, last row / column) of the tap-rectangle table -- so that no
  //      pixel of the patch reads memory in the epilogue --, scale and bias
  const bool has_bias = p.bias != nullptr;
  v4f pre_tab, pre_sc;
  uint2 pre_bs;
  const bool tab_on = tid < 9 * (BN / 4), pre_on = tid < BN / 4;
  const int tab_q = tid % (BN / 4), tab_c = tid / (BN / 4);      // channel quad, class 0..8
  const bool tab_in = n0 + tab_q * 4 < p.K;
  if (tab_on) {
    const int rc = tab_c / 3, cc = tab_c % 3;                       // 0 full, 1 first, 2 last
    const int rlo = rc == 1 ? 1 : 0, rhi = rc == 2 ? 1 : 2, slo = cc == 1 ? 1 : 0, shi = cc == 2 ? 1 : 2;
    const int cls = ((rlo * 3 + rhi) * 3 + slo) * 3 + shi;          // igemm.hip's class index
    const int n = tab_in ? n0 + tab_q * 4 : 0;
    pre_tab = *reinterpret_cast<const v4f*>(p.table + (int64_t)cls * p.K + n);
    if (pre_on) {
      pre_sc = *reinterpret_cast<const v4f*>(p.scale + n);
      if (has_bias) pre_bs = *reinterpret_cast<const uint2*>(p.bias + n);
    }
  }

  // ---- per-lane DMA state ---------------------------------------------------------------------
  // halo piece j of this wave = piece wid + 8 j: PPP halo pixels x CK bytes; lane -> (pixel, 16-B
  // slot); the slot holds source chunk slot ^ swz(pixel)
  uint32_t h_off[G::H_NI];
  int h_c16[G::H_NI];
  bool h_ok[G::H_NI];
#pragma unroll
  for (int j = 0; j < G::H_NI; ++j) {
    const int h = (wid + 8 * j) * G::PPP + lane / G::LPP;
    const int hy = h / HWP, hx = h - hy * HWP;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const int c16 = ((lane % G::LPP) ^ G::swz(h)) << 4;
    h_ok[j] = h < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    h_off[j] = h_ok[j] ? (uint32_t)((iy >> sh) * SW + (ix >> sh)) * (uint32_t)C + c16 : 0u;
    h_c16[j] = c16;
  }
  // weight piece j of this wave = piece wid + 8 j of a filter row's stage: tap s = piece / (pieces
  // per tap tile), PPP output channels x CK bytes
  uint32_t w_off[G::W_NI];
  int w_c16[G::W_NI];
  bool w_ok[G::W_NI];
#pragma unroll
  for (int j = 0; j < G::W_NI; ++j) {
    const int q = wid + 8 * j;
    const int s = q / (BN / G::PPP), row = (q - s * (BN / G::PPP)) * G::PPP + lane / G::LPP;
    const int c16 = ((lane % G::LPP) ^ G::swz(row)) << 4;
    w_ok[j] = q < G::W_PIECES && n0 + row < p.K;
    w_off[j] = w_ok[j] ? (uint32_t)(n0 + row) * (uint32_t)(9 * C) + (uint32_t)(s * C) + c16 : 0u;
    w_c16[j] = c16;
  }
  auto stage_halo = [&](int chunk, int buf) {
    char* dst = smem + G::HALO_OFF + buf * G::HALO_BYTES;
    const int c0 = chunk * CK;
    const bool live = chunk < nch;
#pragma unroll
    for (int j = 0; j < G::H_NI; ++j) {
      const bool ok = live && h_ok[j] && c0 + h_c16[j] < C;
      dma16(ok ? (const void*)(ximg + h_off[j] + c0) : (const void*)zero, dst + (wid + 8 * j) * 1024);
    }
  };
  auto stage_w = [&](int chunk, int r) {       // filter row r of a channel chunk -> stage r
    char* dst = smem + r * G::W_STAGE;
    const int c0 = chunk * CK;
    const bool live = chunk < nch;
    const int8_t* base = p.Wt + (r * 3) * C + c0;
#pragma unroll
    for (int j = 0; j < G::W_NI; ++j) {
      if (j >= G::W_LO && wid >= G::W_REM) continue;     // wave-uniform: this wave has no such piece
      const bool ok = live && w_ok[j] && c0 + w_c16[j] < C;
      dma16(ok ? (const void*)(base + w_off[j]) : (const void*)zero, dst + (wid + 8 * j) * 1024);
    }
  };
  // counted wait for filter row (chunk, r): this wave's pieces of the NEXT row may still fly -- and,
  // at r == 1, the next chunk's halo, issued one step ago behind row (chunk, 1)
  auto wait_row = [&](auto r_c) {
    constexpr int r = decltype(r_c)::value;
    constexpr int extra = r == 1 ? G::H_NI : 0;
    if constexpr (G::W_REM == 0) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G::W_LO + extra) : "memory");
    } else {
      if (wid < G::W_REM) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G::W_NI + extra) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G::W_LO + extra) : "memory");
    }
  };

  // ---- fragment read offsets (loop invariant) ---------------------------------------------------
  // activations: pixel (ty, tx) of the patch under tap (r, s) = halo pixel (ty + r) * HWP + tx + s;
  // this wave's k-step is kg (64 of the chunk's 128 bytes), a lane's chunk lkq of it
  int x_rd[TM][9];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int i = wm * G::WTM + t * 16 + lrow;
    const int hb = (i / TW) * HWP + (i % TW);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int h = hb + (tap / 3) * HWP + (tap % 3);
      x_rd[t][tap] = G::HALO_OFF + h * CK + (((kg * 4 + lkq) ^ G::swz(h)) << 4);
    }
  }
  int w_rd[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int row = wn * G::WTN + tn * 16 + lrow;
    w_rd[tn] = row * CK + (((kg * 4 + lkq) ^ G::swz(row)) << 4);
  }
  v4i acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = v4i{0, 0, 0, 0};

  // ---- main loop: per channel chunk three steps (filter rows), halo buffer = chunk parity ----------
  stage_halo(0, 0);
  stage_w(0, 0);
  stage_w(0, 1);
  auto chunk_body = [&](auto hb_c, int ch) {
    constexpr int HB = decltype(hb_c)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (r == 0) wait_row(std::integral_constant<int, 0>{});
      else if (r == 1) wait_row(std::integral_constant<int, 1>{});
      else wait_row(std::integral_constant<int, 2>{});
      // stage: the next chunk's halo (its buffer was last read a full chunk ago), then the weight
      // row two steps ahead (into the stage read one step ago)
      if (r == 0) stage_halo(ch + 1, HB ^ 1);
      stage_w(ch + (r + 2) / 3, (r + 2) % 3);
      const char* Ws = smem + r * G::W_STAGE;
      const char* Hs = smem + HB * G::HALO_BYTES;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        v4i wf[TN], xf[TM];
#pragma unroll
        for (int t = 0; t < TM; ++t) xf[t] = *reinterpret_cast<const v4i*>(Hs + x_rd[t][r * 3 + s]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          wf[tn] = *reinterpret_cast<const v4i*>(Ws + s * G::W_TAP + w_rd[tn]);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int t = 0; t < TM; ++t)
            acc[tn][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[tn], xf[t], acc[tn][t], 0, 0, 0);
      }
    }
  };
  for (int ch = 0; ch < nch; ch += 2) {
    chunk_body(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < nch) chunk_body(std::integral_constant<int, 1>{}, ch + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-page DMAs staged past the end

  // ---- epilogue --------------------------------------------------------------------------------
  constexpr int PARAM_OFF = G::MAIN_BYTES;
  float* P_TAB = reinterpret_cast<float*>(smem + PARAM_OFF);       // [9][BN]
  float* P_SC = P_TAB + 9 * BN;
  __half* P_BS = reinterpret_cast<__half*>(P_SC + BN);
  if (tab_on) {
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<v4f*>(P_TAB + tab_c * BN + tab_q * 4) = tab_in ? pre_tab : zero4;
    if (pre_on) {
      *reinterpret_cast<v4f*>(P_SC + tab_q * 4) = tab_in ? pre_sc : zero4;
      *reinterpret_cast<uint2*>(P_BS + tab_q * 4) = tab_in && has_bias ? pre_bs : make_uint2(0u, 0u);
    }
  }
  __syncthreads();                                   // every wave is done reading the main buffers
  // k-split: group 1 parks its partial accumulators behind the fp16 tile's area, group 0 adds them
  if constexpr (G::KSPLIT == 2) {
    constexpr int WREGS = TN * TM * 4;
    int* part = reinterpret_cast<int*>(smem + BM * CS_STRIDE) + (wm * WREGS * 64 + lane);
    if (kg != 0) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[((a * TM + b) * 4 + e) * 64] = acc[a][b][e];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][e] += part[((a * TM + b) * 4 + e) * 64];
    }
  }
  char* Cs = smem;
  auto to_tile = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;    // 0: no bias, 1: bias (FMA), 2: bias, mul then add
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int ml = wm * G::WTM + t * 16 + lrow;
      // border class of this output pixel = its valid tap rectangle [rlo, rhi] x [slo, shi]
      // (conv_act_zero_point_propagate.cu:23-51 restated as a table lookup, igemm.hip): one of
      // the nine rows staged in LDS
      const int pp = y0 + ml / TW, qq = x0 + ml % TW;
      const int rc = pp == 0 ? 1 : (pp == p.H - 1 ? 2 : 0), cc = qq == 0 ? 1 : (qq == p.W - 1 ? 2 : 0);
      const float* b0row = P_TAB + (rc * 3 + cc) * BN;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nl = wn * G::WTN + tn * 16 + 4 * lkq;
        v4f b0 = *reinterpret_cast<const v4f*>(b0row + nl);
        b0 = b0 * zpv;                                 // f32(sum of in-image taps) * zp, one rounding
        const v4f sc = *reinterpret_cast<const v4f*>(P_SC + nl);
        v4f bs = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE != 0)
          bs = __builtin_convertvector(*reinterpret_cast<const v4h*>(P_BS + nl), v4f);
        uint32_t packed[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          v2f x = {(float)acc[tn][t][2 * e2], (float)acc[tn][t][2 * e2 + 1]};
          const v2f b0e = {b0[2 * e2], b0[2 * e2 + 1]};
          const v2f sce = {sc[2 * e2], sc[2 * e2 + 1]};
          const v2f bse = {bs[2 * e2], bs[2 * e2 + 1]};
          v2f r;
          x = x - b0e;
          if constexpr (MODE == 0) r = x * sce;
          else if constexpr (MODE == 2) r = x * sce + bse;
          else r = __builtin_elementwise_fma(x, sce, bse);
          asm("" : "+v"(r));       // keep the FP32 rounding (no single-rounding fma_mix fold)
          const v2h h = __builtin_convertvector(r, v2h);
          packed[e2] = *reinterpret_cast<const uint32_t*>(&h);
        }
        *reinterpret_cast<uint2*>(Cs + ml * CS_STRIDE + nl * 2) = make_uint2(packed[0], packed[1]);
      }
    }
  };
  if (kg == 0) {
    if (!has_bias) to_tile(std::integral_constant<int, 0>{});
    else if (!p.unfused) to_tile(std::integral_constant<int, 1>{});
    else to_tile(std::integral_constant<int, 2>{});
  }
  // whole 16-byte row segments out; thread -> (pixel, chunk), chunk fastest
  constexpr int CPRO = BN / 8;
  const bool n8 = (p.K & 7) == 0;
  // the residual operand (conv2: the block input, cold -- another kernel just wrote it): every chunk
  // this thread needs is requested here, at once, and lands under the barrier and the LDS reads below.
  // (Read inside the store loop it cost one dependent memory round trip per iteration: the compiler
  // cannot move a load of `res` across a store to `D`.)
  constexpr int ST_ITERS = (BM * CPRO + G::NTHREADS - 1) / G::NTHREADS;
  static_assert(ST_ITERS <= 10, "residual chunks parked in registers");
  v4i res_late[ST_ITERS];
  const bool res_late_on = p.res != nullptr && n8;
  if (res_late_on) {
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
      const int idx = min(tid + it * G::NTHREADS, BM * CPRO - 1);
      const int row = idx / CPRO, cc = idx - row * CPRO;
      const int n = n0 + cc * 8 < p.K ? n0 + cc * 8 : 0;
      const int64_t pix = ((int64_t)img * p.H + (y0 + row / TW)) * p.W + (x0 + row % TW);
      res_late[it] = *reinterpret_cast<const v4i*>(p.res + (p.res_div == 1 ? pix : (int64_t)img) * p.K + n);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ST_ITERS; ++it) {
    const int idx = tid + it * G::NTHREADS;
    if (idx >= BM * CPRO) break;
    const int row = idx / CPRO, cc = idx - row * CPRO;
    const int n = n0 + cc * 8;
    if (n >= p.K) continue;
    const int64_t pix = ((int64_t)img * p.H + (y0 + row / TW)) * p.W + (x0 + row % TW);
    uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_STRIDE + cc * 16);
    if (p.res != nullptr) {
      const __half* rp = p.res + (p.res_div == 1 ? pix : (int64_t)img) * p.K + n;
      uint32_t rw[4];
      if (n8) {
        const v4i r = res_late[it];
        rw[0] = (uint32_t)r[0]; rw[1] = (uint32_t)r[1]; rw[2] = (uint32_t)r[2]; rw[3] = (uint32_t)r[3];
      } else {
        const uint2 r0 = *reinterpret_cast<const uint2*>(rp);
        rw[0] = r0.x; rw[1] = r0.y; rw[2] = 0; rw[3] = 0;
        if (n + 8 <= p.K) {
          const uint2 r1 = *reinterpret_cast<const uint2*>(rp + 4);
          rw[2] = r1.x; rw[3] = r1.y;
        }
      }
      uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) vw[e] = add_f16x2_(vw[e], rw[e]);
    }
    __half* dst = p.D + pix * p.K + n;
    if (n8) {
      const v4i vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
      // global_store_dwordx4 ... nt, through the builtin: the compiler then counts the store in vmcnt and pads
        // the gfx950 store-data hazard (a store of more than 64 bits reads its data registers up to two wait
        // states after issue).  As inline asm it did neither: a VALU write of the first data register right
        // behind the asm was picked up by the store (tests/test_ops_gpu.py halo cases, round 4).
        __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
    } else {
      *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
      if (n + 8 <= p.K) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(v.z, v.w);
    }
  }
}

template <int TH, int TW, int BN, int CK, int WNG = 1>
int launch_halo(const HaloConvArgs& a, hipStream_t stream) {
  using G = HaloGeom<TH, TW, BN, CK, WNG>;
  static bool seen[64] = {};
  if (const int st = lds_opt_in(reinterpret_cast<const void*>(&conv3x3_halo_kernel<TH, TW, BN, CK, WNG>),
                                G::SMEM, seen))
    return st;
  const int64_t grid = (int64_t)a.NI * (a.H / TH) * (a.W / TW) * ((a.K + BN - 1) / BN);
  if (grid <= 0 || grid > 0x7fffffff) return MIXDQ_ERR_INVALID_ARG;
  conv3x3_halo_kernel<TH, TW, BN, CK, WNG><<<dim3((unsigned)grid), 512, G::SMEM, stream>>>(a);
  return launch_status();
}

}  // namespace

// Patches (tools/bench_gemm.py --conv): the weights are the larger stream (9 x BN x CK bytes per chunk
// against one halo), so the more pixels share them the fewer bytes per output -- 16 x 16 pixels
// (65 KB per 128 x 80 outputs and 128-byte chunk against 113 KB for 8 x 16) as soon as that still
// gives every CU a workgroup (128 x 128 x 320 at batch 1: 21.2 vs 28.8 us; every layer from batch 4
// on); otherwise the patch that fills the chip: 8 x 16, or 8 x 8 for the 32 x 32 layers at batch 1.
int halo_conv_select(int NI, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  if (R != 3 || S != 3 || stride != 1 || pad != 1 || NI <= 0) return 0;
  if (C % 64 != 0 || K % 4 != 0 || H % 8 != 0 || W % 8 != 0) return 0;
  // 32-bit per-lane offsets inside one image / the weight tensor
  if ((int64_t)H * W * C >= (1ll << 32) || (int64_t)K * 9 * C >= (1ll << 32)) return 0;
  const int64_t tiles_n = (K + 79) / 80;
  // 160 channels per workgroup on 4 x 2 waves of 64 pixels x 80 channels: half the halo bytes and 0.64 of
  // the LDS fragment reads per MAC of tile 92, when that still gives every CU a workgroup
  if (H % 16 == 0 && W % 16 == 0 && K % 160 == 0 && (int64_t)NI * (H / 16) * (W / 16) * (K / 160) >= kNumCU)
    return 93;
  if (H % 16 == 0 && W % 16 == 0 && (int64_t)NI * (H / 16) * (W / 16) * tiles_n >= kNumCU) return 92;
  if (W % 16 == 0 && (int64_t)NI * (H / 8) * (W / 16) * tiles_n >= kNumCU) return 90;
  return 91;
}

int halo_conv_launch(const HaloConvArgs& a, int tile, hipStream_t stream) {
  switch (tile) {
    case 90: return launch_halo<8, 16, 80, 128>(a, stream);
    case 91: return launch_halo<8, 8, 80, 128>(a, stream);
    case 92: return launch_halo<16, 16, 80, 64>(a, stream);
    case 93: return launch_halo<16, 16, 160, 64, 2>(a, stream);
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

}  // namespace mixdq
