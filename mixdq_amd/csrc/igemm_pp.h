// Persistent form of the four-phase 256 x 256 x 128 INT8 GEMM (round 6).  Included by igemm.hip behind
// igemm_kernel.h: same operands, same tile arithmetic, same epilogue arithmetic -- the SAME BITS as
// igemm_kernel<256, 256, 128, 2, ..., PHASED> -- launched as ONE workgroup per CU that walks its share of the tile
// sequence.  What the tile loop buys over one workgroup per tile:
//   * the NEXT tile's first K-tile (64 KB: A0, B0, B1, A1) is requested by LDS-DMA during the LAST K-tile of the
//     current one -- the slot where the one-tile kernel stages dummy reads to keep its counted waits uniform -- into
//     the stage buffer the epilogue does not use, so it lands under the accumulator -> fp16 / GEGLU / store passes
//     (2.5-3 + 3-6 + 1 us on this tile) instead of in front of an idle MFMA pipe: no prologue,
//     no first-K-tile wait, no per-workgroup entry (argument block, tile map, staging offsets) from the second
//     tile on;
//   * nothing is exchanged between workgroups (no seam of MI355X_MICROARCH.md's price list is paid).
// What it cannot do, and why: run tile i's epilogue UNDER tile i + 1's main loop.  A 256 x 256
// INT32 tile is 128 accumulator registers per lane at two waves per SIMD (232 of 256 with the fragments); a second
// resident tile, or the 64 packed-fp16 registers of a pending one, do not fit, and the GELU table (76 KB) cannot
// sit in LDS beside two 64 KB stage buffers.
//
// (Measured and dropped, round 6: the next tile's SECOND K-tile requested from the GEGLU epilogue as well -- all table
//  lookups first, the table laid over stage L and the staging tile in the extra block, so that stage L is free ~2 us
//  before the main loop; bit-identical, and SLOWER: (8192, 10240, 1280) 107.7 -> 110.5 us, (32768, 5120, 640) 138 -> 145.5
//  (profiles/r06_persistent_early_k1_ab.txt): with the lookups in front the store passes no longer run beside lookup
//  arithmetic, and the early DMA shares the CU's memory pipe with the stores.  hipcc also put `s_waitcnt vmcnt(0)` in
//  front of every compiler-generated LDS read behind that DMA; those reads had to become inline asm.)
//
// LDS (all 160 KiB): [params 3 K][stage 0 64 K][extra 29 K][stage 1 64 K].  The epilogue of a tile whose LAST K-tile
// sat in stage L works in the 93 KB that contain stage L and the extra block -- [3 K, 96 K) or [67 K, 160 K):
// contiguous either way -- while stage L ^ 1 receives the next tile's first K-tile.  93 KB = the GELU table (76 KB)
// + an INT8 staging tile of 128 rows x 136 B, exactly; or an fp16 staging tile of 128 rows x 528 B: the output
// leaves in two passes of 128 rows.  The per-channel epilogue vectors arrive by LDS-DMA too (three 1-KiB pieces per
// tile, requested during the tile's first K-tile; columns past N read the zero page).
#pragma once

namespace mixdq {
namespace {

constexpr int kPpParam = 3072, kPpStage = 65536, kPpExtra = 160 * 1024 - kPpParam - 2 * kPpStage;
constexpr int kPpS0 = kPpParam, kPpX = kPpS0 + kPpStage, kPpS1 = kPpX + kPpExtra;
constexpr int kPpRegion = kPpStage + kPpExtra;
constexpr int kPpQS = 136;                         // INT8 staging row stride (bytes): 128 + 8
constexpr int kPpCS = 256 * 2 + 16;                // fp16 staging row stride (bytes)
static_assert(kPpS1 + kPpStage == 160 * 1024, "all of the CU's LDS");
static_assert(128 * kPpQS + kGeluTabBytes <= kPpRegion, "INT8 staging + GELU table fit the epilogue region");
static_assert(128 * kPpCS <= kPpRegion, "fp16 staging fits the epilogue region");

template <bool GEGLU>
__global__ __launch_bounds__(512, 2) void igemm_pp_kernel(MIXDQ_IGEMM_HEAD_PARAMS const IgemmParams p_in) {
  constexpr int BM = 256, BN = 256, BK = 128, A_STAGE = BM * BK;
  constexpr int WTM = 128, WTN = 64, TM = 8, TN = 4, CPS = 4;
  MIXDQ_ARGS_NOW(p_in.bias, p_in.D, p_in.Dq, p_in.res, p_in.res_div, p_in.unfused, p_in.g_sinv, p_in.g_zp);
  IgemmParams p = p_in;
  MIXDQ_IGEMM_HEAD_TAKE(p);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int lrow = lane & 15, lkq = lane >> 4;
  MIXDQ_STAMP_AT(0);

  // ---- this workgroup's share of the tile sequence: XCD x (blocks b and b + 8 share one: an observation, used for
  //      speed only) owns the same contiguous run of the sequence as in igemm_kernel's map; its G / 8 resident
  //      workgroups take the run's tiles round-robin, i.e. in the order the dispatcher would have handed them out
  const int nwg = p.tiles_m * p.tiles_n;
  const int G = (int)gridDim.x, w8 = G / kNumXCD;
  const int xcd = (int)blockIdx.x % kNumXCD, slot = (int)blockIdx.x / kNumXCD;
  const int q8 = nwg / kNumXCD, r8 = nwg % kNumXCD;
  const int run0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int run1 = run0 + q8 + (xcd < r8 ? 1 : 0);
  int seq = run0 + slot;
  if (seq >= run1) return;
  const int GM = p.gm, per_group = GM * p.tiles_n;
  auto tile_of = [&](int s, int& tm_, int& tn_) {
    const int group = s / per_group, first_m = group * GM;
    const int gsz = min(GM, p.tiles_m - first_m), rem = s - group * per_group;
    tn_ = rem / gsz;
    tm_ = first_m + (rem - tn_ * gsz);
  };
  const int Ktot = p.Ktot;
  const int nk = Ktot / BK;
  const char* zero = reinterpret_cast<const char*>(&g_zero16);

  // ---- per-tile staging offsets (as igemm_kernel's PHASED path): unit 0 A0, 1 B0, 2 B1, 3 A1, two pieces per wave
  unsigned ph_a[2][2], ph_b[2][2];
  // (rows in 32 bits: M * K < 2^32.  The lane index passes through an empty asm so that the compiler recomputes these
  //  few operations per tile instead of keeping sixteen loop-invariant values alive across the main loop -- spilled,
  //  their reloads put `s_waitcnt vmcnt(0)` into the K loop: scratch loads count like the DMAs.)
  const uint32_t Mlast = (uint32_t)(p.M - 1), Nlast = (uint32_t)(p.N - 1);
  auto set_offsets = [&](uint32_t m0, uint32_t n0) {
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = wid * 2 + j;
        const int ra = (q >> 3) * 128 + h * 64 + (q & 7) * 8 + (l >> 3);
        const int rb = (q >> 2) * 64 + h * 32 + (q & 3) * 8 + (l >> 3);
        ph_a[h][j] = min(m0 + (uint32_t)ra, Mlast) * (uint32_t)Ktot + (((l & 7) ^ swz<BK>(ra)) << 4);
        ph_b[h][j] = min(n0 + (uint32_t)rb, Nlast) * (uint32_t)Ktot + (((l & 7) ^ swz<BK>(rb)) << 4);
      }
  };
  auto stage_unit = [&](int sbase, int kk, int unit) {   // sbase: kPpS0 / kPpS1; unit: compile-time at every call
    const int kk_u = __builtin_amdgcn_readfirstlane(kk);
    char* S = smem + sbase;
    const bool is_a = unit == 0 || unit == 3;
    const int h = unit >= 2 ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wid * 2 + j;
      if (is_a)
        glds16(p.A + kk_u + ph_a[h][j], S + ((q >> 3) * 128 + h * 64 + (q & 7) * 8) * BK);
      else
        glds16(p.Wt + kk_u + ph_b[h][j], S + A_STAGE + ((q >> 2) * 64 + h * 32 + (q & 3) * 8) * BK);
    }
  };
  // the per-channel vectors of the tile at column n0 -> the parameter block: bias0 (wave 0), scale (wave 1), bias
  // (wave 2, 512 B: the lower half-wave); a 16-byte piece wholly past N reads the zero page (N % 4 == 0)
  float* P_B0 = reinterpret_cast<float*>(smem);
  float* P_SC = P_B0 + BN;
  __half* P_BS = reinterpret_cast<__half*>(P_SC + BN);
  const bool has_bias = p.bias != nullptr;
  auto stage_params = [&](int n0) {
    if (wid == 0) {
      const int n = n0 + lane * 4;
      glds16(n < p.N ? (const void*)(p.bias0 + n) : (const void*)zero, reinterpret_cast<char*>(P_B0));
    } else if (wid == 1) {
      const int n = n0 + lane * 4;
      glds16(n < p.N ? (const void*)(p.scale + n) : (const void*)zero, reinterpret_cast<char*>(P_SC));
    } else if (wid == 2 && has_bias) {
      // 256 halves = 512 B = 32 lanes x 16 B; the upper half-wave re-reads the zero page into the slack behind
      const int n = n0 + lane * 8;
      glds16(lane < 32 && n < p.N ? (const void*)(p.bias + n) : (const void*)zero, reinterpret_cast<char*>(P_BS));
    }
  };

  int tile_m, tile_n;
  tile_of(seq, tile_m, tile_n);
  uint32_t m0 = (uint32_t)tile_m * BM;      // (32 bits: M * K < 2^32)
  int n0 = tile_n * BN;
  set_offsets(m0, (uint32_t)n0);
  stage_unit(kPpS0, 0, 0);
  stage_unit(kPpS0, 0, 1);
  stage_params(n0);
  stage_unit(kPpS0, 0, 2);
  stage_unit(kPpS0, 0, 3);
  MIXDQ_STAMP_AT(1);

  // ---- fragment reads and the four phases of a K-tile: igemm_kernel's PHASED path, verbatim
  const int a_lane = (wm * WTM + lrow) * BK, b_lane = A_STAGE + (wn * WTN + lrow) * BK;
  int fr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fr[ks] = ((ks * CPS + lkq) ^ swz<BK>(lrow)) << 4;
  v4i af[4][2], bf[2][2];
  v4i acc[TN][TM];
  auto read_a = [&](const char* S0, int mh) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[t][ks] = *reinterpret_cast<const v4i*>(S0 + a_lane + (mh * 4 + t) * 16 * BK + fr[ks]);
  };
  auto read_b = [&](const char* S0, int nh) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        bf[t][ks] = *reinterpret_cast<const v4i*>(S0 + b_lane + (nh * 2 + t) * 16 * BK + fr[ks]);
  };
  auto quadrant = [&](int mh, int nh) {
    asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[nh * 2 + tn][mh * 4 + tm] =
              __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[tn][ks], af[tm][ks], acc[nh * 2 + tn][mh * 4 + tm], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
  };
#define MIXDQ_PP_WAIT() asm volatile("s_waitcnt vmcnt(4)" ::: "memory")

  const bool unfused = p.unfused != 0;
  const int mode = !has_bias ? 0 : !unfused ? 1 : 2;
  float g_sinv = 0.f, g_zpq = 0.f;
  if constexpr (GEGLU) {      // (scalar loads: constant address space)
    g_sinv = *(const __attribute__((address_space(4))) float*)p.g_sinv;
    g_zpq = *(const __attribute__((address_space(4))) float*)p.g_zp;
  }

  int par = 0;                 // the stage buffer that holds K-tile 0 of the current tile
  bool first = true;
  for (;;) {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = v4i{0, 0, 0, 0};
    // K-tile 0 is in LDS and visible to every wave: the first tile waits for its A0, B0 (and the parameter
    // pieces, requested between B0 and B1) here; a later tile's first K-tile was drained by every wave in front of
    // the previous epilogue's stores, and this barrier is also the one that frees that epilogue's LDS
    if (first) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (!first) MIXDQ_STAMP_AT(10);
    else MIXDQ_STAMP_AT(2);
    if (wid >= 4) asm volatile("s_barrier" ::: "memory");   // the second wave group: one barrier behind
    int next_seq = seq + w8;
    bool has_next = next_seq < run1;
    uint32_t m0n = m0;
    int n0n = n0;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = (par + kt) & 1;
      const char* S0 = smem + (cur ? kPpS1 : kPpS0);
      const int NB = cur ? kPpS0 : kPpS1;
      int nkk = (kt + 1) * BK;
      if (kt == nk - 1) {      // what is staged from here on is the NEXT tile's first K-tile (none: this tile's, unused)
        nkk = 0;
        if (has_next) {
          int tm_, tn_;
          tile_of(next_seq, tm_, tn_);
          m0n = (uint32_t)tm_ * BM;
          n0n = tn_ * BN;
          set_offsets(m0n, (uint32_t)n0n);
        }
      }
      const bool waits = first || kt > 0;   // (a later tile's K-tile 0 has landed as a whole, see above; its first two
      read_b(S0, 0); __builtin_amdgcn_sched_barrier(0); read_a(S0, 0);   //  waits would only wait for the last stores)
      stage_unit(NB, nkk, 0);
      if (waits) MIXDQ_PP_WAIT();                           // B1 of this K-tile
      quadrant(0, 0);
      read_b(S0, 1);
      stage_unit(NB, nkk, 1);
      if (waits) MIXDQ_PP_WAIT();                           // A1 of this K-tile
      quadrant(0, 1);
      read_a(S0, 1);
      stage_unit(NB, nkk, 2);
      quadrant(1, 1);
      read_b(S0, 0);
      stage_unit(NB, nkk, 3);
      MIXDQ_PP_WAIT();                                      // A0, B0 of the next K-tile
      // this tile's epilogue vectors: the previous tile's were last read in front of this tile's first barrier
      if (kt == 0 && !first) stage_params(n0);
      quadrant(1, 0);
    }
    if (wid < 4) asm volatile("s_barrier" ::: "memory");    // the groups meet again
    if (first) MIXDQ_STAMP_AT(3);
    else MIXDQ_STAMP_AT(11);
    const int L = (par + nk - 1) & 1;                        // the stage that is free now
    char* R = smem + (L ? kPpX : kPpS0);                     // the epilogue's 93 KB
    // every index of the epilogue derives from THIS copy of the thread index: opaque, so that none of it is computed
    // ahead of the tile loop and kept alive (or spilled: a reload behind the main loop waits vmcnt(0) and would drain
    // the next tile's first K-tile) across the main loop
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, lrow_e = lane_e & 15, lkq_e = lane_e >> 4;

    if constexpr (GEGLU) {
      // ---- GEMM + GEGLU + quantize in registers (igemm_kernel's epilogue on this tile: value | gate groups of 16,
      //      GELU by table), the INT8 tile staged and stored in two passes of 128 rows
      const char* Tb = R + 128 * kPpQS;
      {
        constexpr int PIECES = kGeluTabBytes / 1024;
        const char* src = reinterpret_cast<const char*>(g_gelu_tab);
#pragma unroll
        for (int j = 0; j < (PIECES + 7) / 8; ++j) {
          const int q = wid + 8 * j;
          if (q < PIECES) glds16(src + q * 1024 + lane_e * 16, R + 128 * kPpQS + q * 1024);
        }
      }
      auto vcol = [&](int oq) { return wn * WTN + oq * 32 + 4 * lkq_e; };
      uint2 hv[TM][2], hg[TM][2];
      auto to_regs = [&](auto mode_c, int tm) {
        constexpr int MODE = decltype(mode_c)::value;
        v4f b0[4], sc[4];
        v4h bsh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nl = vcol(j >> 1) + 16 * (j & 1);
          b0[j] = *reinterpret_cast<const v4f*>(P_B0 + nl);
          sc[j] = *reinterpret_cast<const v4f*>(P_SC + nl);
          if constexpr (MODE != 0) bsh[j] = *reinterpret_cast<const v4h*>(P_BS + nl);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int oq = j >> 1, half = j & 1;
          const int tn = 2 * oq + half;
          v4f bs = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MODE != 0) bs = __builtin_convertvector(bsh[j], v4f);
          uint32_t packed[2];
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            v2f x = {(float)acc[tn][tm][2 * e2], (float)acc[tn][tm][2 * e2 + 1]};
            const v2f b0e = {b0[j][2 * e2], b0[j][2 * e2 + 1]};
            const v2f sce = {sc[j][2 * e2], sc[j][2 * e2 + 1]};
            const v2f bse = {bs[2 * e2], bs[2 * e2 + 1]};
            v2f r;
            x = x - b0e;
            if constexpr (MODE == 0) r = x * sce;
            else if constexpr (MODE == 2) r = x * sce + bse;
            else r = __builtin_elementwise_fma(x, sce, bse);
            asm("" : "+v"(r));
            const v2h h = __builtin_convertvector(r, v2h);
            packed[e2] = *reinterpret_cast<const uint32_t*>(&h);
          }
          if (half == 0) hv[tm][oq] = make_uint2(packed[0], packed[1]);
          else hg[tm][oq] = make_uint2(packed[0], packed[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        if (mode == 0) to_regs(std::integral_constant<int, 0>{}, tm);
        else if (mode == 1) to_regs(std::integral_constant<int, 1>{}, tm);
        else to_regs(std::integral_constant<int, 2>{}, tm);
      }
      if (first) MIXDQ_STAMP_AT(5);
      else MIXDQ_STAMP_AT(12);
      // the table has landed -- and with it everything older: the next tile's first K-tile, whole
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (first) MIXDQ_STAMP_AT(6);
      else MIXDQ_STAMP_AT(13);
      const int Dh = p.N >> 1;
      const bool al16 = ((uintptr_t)p.Dq & 15) == 0;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int tm = pass * 4 + t4;
          const int sl = wm * 64 + t4 * 16 + lrow_e;            // staging row: 64 rows of each row half per pass
          uint32_t pk[2];
          auto quads = [&](auto how_c, auto unf_c) {
            constexpr int HOW = decltype(how_c)::value;
            constexpr bool UNF = decltype(unf_c)::value;
#pragma unroll
            for (int oq = 0; oq < 2; ++oq) pk[oq] = geglu_quad<HOW, UNF>(hv[tm][oq], hg[tm][oq], Tb, g_sinv, g_zpq);
          };
          const bool far = geglu_any_far(hg[tm][0].x, hg[tm][0].y, hg[tm][1].x, hg[tm][1].y);
          if (__builtin_amdgcn_ballot_w64(far) == 0) {
            if (unfused) quads(std::integral_constant<int, 0>{}, std::true_type{});
            else quads(std::integral_constant<int, 0>{}, std::false_type{});
          } else {
            if (unfused) quads(std::integral_constant<int, 1>{}, std::true_type{});
            else quads(std::integral_constant<int, 1>{}, std::false_type{});
          }
#pragma unroll
          for (int oq = 0; oq < 2; ++oq) {
            const int v = vcol(oq);
            *reinterpret_cast<uint32_t*>(R + sl * kPpQS + ((v & ~31) >> 1) + (v & 15)) = pk[oq];
          }
        }
        // (raw barriers in the pass loop: __syncthreads() also waits vmcnt(0), i.e. for the acknowledgement of the
        //  stores the previous pass has just issued -- ~1 us per tile of nothing; what is ordered here is LDS traffic)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // 128 staged rows x 8 sixteen-byte chunks, two per thread; whole 128-byte rows leave per 8 lanes
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int idx = tid_e + it * 512, srow = idx >> 3, cc = idx & 7;
          const int64_t m = (int64_t)m0 + (srow >> 6) * 128 + pass * 64 + (srow & 63);
          if (m >= p.M || n0 + 32 * cc >= p.N) continue;
          const uint2 lo = *reinterpret_cast<const uint2*>(R + srow * kPpQS + cc * 16);
          const uint2 hi = *reinterpret_cast<const uint2*>(R + srow * kPpQS + cc * 16 + 8);
          int8_t* dst = p.Dq + m * Dh + (n0 >> 1) + cc * 16;
          if (al16) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
          } else {
            *reinterpret_cast<uint2*>(dst) = lo;
            *reinterpret_cast<uint2*>(dst + 8) = hi;
          }
        }
        if (pass == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (pass 1: the next tile's first barrier, or the exit)
        if (pass == 0 && !first) MIXDQ_STAMP_AT(15);
      }
    } else {
      // ---- plain epilogue: accumulators -> fp16 tile (igemm_kernel's to_tile arithmetic) -> whole-row stores,
      //      two passes of 128 rows (64 of each row half); optional residual, requested in front of each pass
      const bool res_on = p.res != nullptr;
      const bool res_full = p.res_div == 1;
      auto to_tile = [&](auto mode_c, int pass) {
        constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          const int tm = pass * 4 + t4;
          const int sl = wm * 64 + t4 * 16 + lrow_e;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const int nl = wn * WTN + tn * 16 + 4 * lkq_e;
            const v4f b0 = *reinterpret_cast<const v4f*>(P_B0 + nl);
            const v4f sc = *reinterpret_cast<const v4f*>(P_SC + nl);
            v4f bs = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE != 0) bs = __builtin_convertvector(*reinterpret_cast<const v4h*>(P_BS + nl), v4f);
            uint32_t packed[2];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              v2f x = {(float)acc[tn][tm][2 * e2], (float)acc[tn][tm][2 * e2 + 1]};
              const v2f b0e = {b0[2 * e2], b0[2 * e2 + 1]};
              const v2f sce = {sc[2 * e2], sc[2 * e2 + 1]};
              const v2f bse = {bs[2 * e2], bs[2 * e2 + 1]};
              v2f r;
              x = x - b0e;
              if constexpr (MODE == 0) r = x * sce;
              else if constexpr (MODE == 2) r = x * sce + bse;
              else r = __builtin_elementwise_fma(x, sce, bse);
              asm("" : "+v"(r));
              const v2h h = __builtin_convertvector(r, v2h);
              packed[e2] = *reinterpret_cast<const uint32_t*>(&h);
            }
            *reinterpret_cast<uint2*>(R + sl * kPpCS + nl * 2) = make_uint2(packed[0], packed[1]);
            if (tn == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        // thread -> (staged row, 16-byte chunk): 128 rows x 32 chunks, eight per thread, chunk-fastest
        v4i rq[8];
        if (res_on) {
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int idx = tid_e + it * 512, srow = idx >> 5, cc = idx & 31;
            const int64_t m = min((int64_t)m0 + (srow >> 6) * 128 + pass * 64 + (srow & 63), p.M - 1);
            const int n = n0 + cc * 8 < p.N ? n0 + cc * 8 : 0;
            rq[it] = *reinterpret_cast<const v4i*>(p.res + (res_full ? m : m / p.res_div) * p.N + n);
          }
        }
        if (mode == 0) to_tile(std::integral_constant<int, 0>{}, pass);
        else if (mode == 1) to_tile(std::integral_constant<int, 1>{}, pass);
        else to_tile(std::integral_constant<int, 2>{}, pass);
        // everything older than the stores has landed: the residual chunks and the next tile's first K-tile, whole
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (first && pass == 0) MIXDQ_STAMP_AT(6);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int idx = tid_e + it * 512, srow = idx >> 5, cc = idx & 31;
          const int64_t m = (int64_t)m0 + (srow >> 6) * 128 + pass * 64 + (srow & 63);
          const int n = n0 + cc * 8;
          if (m >= p.M || n >= p.N) continue;
          uint4 v = *reinterpret_cast<const uint4*>(R + srow * kPpCS + cc * 16);
          if (res_on) {
            uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
            for (int e = 0; e < 4; ++e) vw[e] = add_f16x2(vw[e], (uint32_t)rq[it][e]);
          }
          const v4i vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
          __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(p.D + m * p.N + n));
        }
        if (pass == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    if (first) MIXDQ_STAMP_AT(7);
    else MIXDQ_STAMP_AT(14);
    if (!has_next) break;
    par = (par + nk) & 1;
    first = false;
    seq = next_seq;
    m0 = m0n;
    n0 = n0n;
  }
#undef MIXDQ_PP_WAIT
}

// One workgroup per CU (the device's real count, rounded down to whole XCD rounds), never more than there are tiles.
// MIXDQ_IGEMM_PERSIST_WGS=<n> caps the grid (tests: several tiles per workgroup on small problems).
template <bool GEGLU>
int launch_pp(IgemmParams& p, hipStream_t stream) {
  static bool seen[64] = {};
  if (const int st = lds_opt_in(reinterpret_cast<const void*>(&igemm_pp_kernel<GEGLU>), 160 * 1024, seen)) return st;
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MIXDQ_ERR_LAUNCH;
    cus[dev] = n >= kNumXCD ? n / kNumXCD * kNumXCD : kNumXCD;
  }
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  p.gm = tile_map_gm();
  const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffff || p.tiles_m >= (1 << 24)) return MIXDQ_ERR_INVALID_ARG;
  int grid = cus[dev];
  if (const char* e = getenv("MIXDQ_IGEMM_PERSIST_WGS")) {
    const int cap = atoi(e) / kNumXCD * kNumXCD;
    if (cap >= kNumXCD && cap < grid) grid = cap;
  }
  if (tiles < grid) grid = (int)((tiles + kNumXCD - 1) / kNumXCD * kNumXCD);
  igemm_pp_kernel<GEGLU><<<dim3((unsigned)grid), 512, 160 * 1024, stream>>>(MIXDQ_IGEMM_HEAD_ARGS(p) p);
  return launch_status();
}

// Is the persistent form (configuration 71) what the automatic choice takes instead of the one-tile-per-workgroup
// four-phase kernel (70)?  Where a CU has more than one tile -- else there is nothing to overlap.
// MIXDQ_IGEMM_PERSIST=0: never (A/B runs).
inline bool pp_auto(int64_t M, int N) {
  static const int on = [] { const char* e = getenv("MIXDQ_IGEMM_PERSIST"); return e ? atoi(e) : 1; }();
  return on && ((M + 255) / 256) * (int64_t)((N + 255) / 256) > kNumCU;
}

// Does the persistent kernel take this launch at all?  The four-phase tile's own range (Linear fast path) with
// identity output rows and at least two K-tiles; anything else runs configuration 70.
inline bool pp_ok(const IgemmParams& p) {
  if (p.Ktot % 128 != 0 || p.Ktot < 256 || p.grp_rows > 0 || p.table != nullptr || p.groups != nullptr) return false;
  if ((uint64_t)p.M * (uint64_t)p.Ktot >= (1ull << 32) || (uint64_t)p.N * (uint64_t)p.Ktot >= (1ull << 32)) return false;
  if (p.Dq == nullptr && (p.N & 7) != 0) return false;       // fp16 rows leave in 16-byte pieces
  return true;
}

}  // namespace
}  // namespace mixdq
