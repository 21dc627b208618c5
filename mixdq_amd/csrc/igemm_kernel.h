// The INT8 (and FP16) GEMM / implicit-GEMM conv kernel template of csrc/igemm.hip, in a header so that its
// instantiation families compile as separate translation units (igemm.hip: the INT8-activation launches;
// igemm_aq.hip: the launches that quantize an FP16 activation operand in their staging path).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "attn_core.h"
#include "iconv.h"
#include "../../include/mixdq_math.h"


// MIXDQ_ABLATE (diagnostic builds only, tools/ablate.sh): 1 = no MFMA, 2 = no LDS fragment reads,
// 3 = no LDS-DMA in the main loop, 4 = no output stores, 5 = no counted DMA waits in the phased loops.
// Results are garbage; the timing shows what the loop waits for.
#define NWAVES_OF(WM, WN, KSPLIT) ((WM) * (WN) * (KSPLIT))
#ifndef MIXDQ_ABLATE
#define MIXDQ_ABLATE 0
#endif

// MIXDQ_STAMP (diagnostic builds only, tools/stamp_build.sh): every wave of every workgroup records the
// shader clock (s_memtime) at the phase boundaries of igemm_kernel into a buffer registered with
// mixdq_debug_stamps(); tools/stamp_report.py turns them into a per-phase time line.
#ifndef MIXDQ_STAMP
#define MIXDQ_STAMP 0
#endif

namespace mixdq {
namespace {

__device__ uint4 g_zero16;   // the zero page (device globals are zero-initialised)
#if MIXDQ_STAMP
__device__ unsigned long long g_stamps;      // address of [workgroup][wave 0..15][16] uint64, or 0
#define MIXDQ_STAMP_AT(slot)                                                                      \
  do {                                                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                   \
    const unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                               \
    const unsigned long long a_ = g_stamps;                                                       \
    if (a_ != 0 && lane == 0) {                                                                   \
      auto sp_ = (__attribute__((address_space(1))) unsigned long long*)a_ +                      \
                 ((size_t)blockIdx.x * 16 + wid) * 16;                                            \
      sp_[slot] = t_;                                                                             \
      if ((slot) == 0) sp_[8] = r_;                                                               \
      if ((slot) == 7) sp_[9] = r_;                                                               \
    }                                                                                             \
  } while (0)
#else
#define MIXDQ_STAMP_AT(slot) do {} while (0)
#endif

struct IgemmParams {
  const int8_t* A;       // activations: [M,Ktot] (linear) or [N,H,W,C] (conv)
  const int8_t* Wt;      // weights [N, Ktot]  (conv: [K,R,S,C])
  const float* bias0;    // [N] (null in table mode)
  const float* scale;    // [N]
  const __half* bias;    // [N] or null
  const float* table;    // [ncls][N] tap-rectangle sums, or null
  const float* zp;       // device scalar (table mode)
  __half* D;             // [M,N]
  int64_t M;
  int N, Ktot;
  int H, W, C, R, S, P, Q, stride, pad;   // conv geometry
  int grp_rows, grp_stride, grp_off;      // output row map (grp_rows <= 0: identity)
  const __half* res;     // optional residual added AFTER the fp16 rounding of the epilogue:
  int64_t res_div;       //   D = f16(f32(f16(epilogue)) + f32(res[(m / res_div) * N + n]))
  int tiles_m, tiles_n;
  int gm;                // m-tiles per super-row of the tile map (8; MIXDQ_IGEMM_GM overrides it for A/B runs)
  int unfused;
  // GEGLU epilogue (Dq != null): the N = 2D output columns are value|gate groups of 16
  // ([v 0..15 | g 0..15 | v 16..31 | ...], weight rows pre-interleaved by the host); the tile is
  // reduced to int8 q(f16(f16(v) * f16(gelu(f16(g))))) [M, D] -- ff.net.2's operand -- instead of D.
  int8_t* Dq;
  const float* g_sinv;
  const float* g_zp;
  // Grouped launch (groups != null): blockIdx.y selects a member; its weights, epilogue vectors,
  // output and N replace the fields above (A, M, K, the row map and the flags are shared).
  const mixdq_gemm_group* groups;
  int ngroups_launch;    // host side only: gridDim.y
  // Cross-attention epilogue (ATT kernels; att_out != null): the tile's fp16 result is to_q's
  // output for 64 query rows x two heads; instead of being stored it is multiplied against the
  // (<= 128) keys / values of those heads, and the attention output leaves as to_out.0's INT8
  // operand (or fp16 when att_sinv is null).
  const __half* att_k; const __half* att_v;    // [B, tkv, N] fp16 (column slices allowed)
  int64_t att_k_bs, att_v_bs;                  // batch strides, elements
  int att_k_rs, att_v_rs, att_tkv, att_tq;     // row strides (elements), keys, query rows per image
  float att_scale_log2;
  void* att_out;                               // [M, N] int8 or fp16
  const float* att_sinv; const float* att_zp;
  // AQ kernels (the activation operand arrives as FP16 and is quantized in the staging path): A is
  // fp16 [M rows][a_ld elements per row] (K <= a_ld: a column slice of a wider tensor is a base pointer
  // and its row stride), quantized with the device scalars a_sinv / a_zp exactly as
  // mixdq_quantize_f16_i8 would; a_rowmap != 0: row m of the operand is row
  // (m / grp_rows) * grp_stride + grp_off + m % grp_rows of the tensor (the BOS slice x[:, 1:, :]).
  const float* a_sinv; const float* a_zp;
  int64_t a_ld;
  int a_rowmap;
  // LNQ kernels: LayerNorm over the N columns of the FINAL output rows (epilogue + residual) and up to three
  // quantizers of the normalised FP16 value -- the consumer layers' INT8 operands -- in the GEMM's own launch.
  // The column tiles of a row block exchange per-16-column partials through `ln_part` and meet at the
  // arrival counter `ln_cnt[2 * tile_m]` (all of them co-resident: the launcher guarantees it).
  const __half* ln_gamma; const __half* ln_beta;
  float ln_eps;
  int ln_nq;
  const float* ln_sinv[3]; const float* ln_zp[3];
  int8_t* ln_q[3];
  __half* ln_h;          // optional FP16 copy of the normalised rows
  float* ln_part;        // [M][N / BN] 16-byte records: {sum, centred sum of squares, launch tag, 0} of each unit
  int* ln_cnt;           // [0]: epoch (launches completed on this workspace), [1]: departures of the running one,
                         // [2]: sticky error word (the tag of a launch in which a record never arrived; 0 = none)
  int ln_local;          // 1: row blocks are laid out XCD by XCD and the records are first looked for in that L2
};

template <int BK>
__device__ __forceinline__ int swz(int row) {
  // 256-B LDS bank row holds 4 (BK=64), 2 (BK=128) or 1 (BK=256) tile rows; XOR so that the 16 lanes of a
  // ds_read_b128 group (rows r..r+3, r+12.., r+20..) land in 16 distinct 16-byte slots.
  return BK == 64 ? ((row >> 2) & 3) : BK == 128 ? ((row >> 1) & 7) : (row & 15);
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- GELU by table ------------------------------------------------------------------------------
// The GEGLU epilogue evaluates GELU on an FP16 gate and rounds the result to FP16: a function of 16
// bits.  Its arithmetic (include/mixdq_math.h: ~75 FP32 operations per element with both erf branches
// taken by every wave) is 5.4-8.6 us of the 26 us (1024, 10240, 1280) launch, the largest of the
// step (tools/stamp_report.py) -- VALU-bound, and packed FP32 buys nothing on CDNA4 (v_pk_fma_f32
// issues at half the rate of v_fma_f32).  The tiles that have a CU to themselves anyway (LDS > 80 KB)
// look the value up instead (the table sits in the K-tile stage buffers, free by then, behind the
// INT8 output tile): f16(gelu(g)) for every |g| < 16 (38 912
// entries, 76 KB; beyond: g, or -0 / NaN as the specification gives), built ONCE per device by the
// specification itself (gelu_table_init_kernel), copied into LDS behind the main loop while the
// accumulators are converted, read with one ds_read_u16 per element.  Bit-identical by construction.
#ifndef MIXDQ_GELU_TAB_MAG
#define MIXDQ_GELU_TAB_MAG 0x4c00   // (0x4800, |g| < 8: batch-1 step 11.62 ms against 11.55 with this, same box)
#endif
constexpr int kGeluTabMag = MIXDQ_GELU_TAB_MAG;             // |g| < 16.0
constexpr int kGeluTabBytes = 2 * kGeluTabMag * 2;          // two signs x 2 bytes
__device__ uint16_t g_gelu_tab[2 * kGeluTabMag];

__global__ __launch_bounds__(256) void gelu_table_init_kernel() {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * kGeluTabMag) return;
  const unsigned short bits = (unsigned short)((i >= kGeluTabMag ? 0x8000 : 0) | (i % kGeluTabMag));
  __half_raw r;
  r.x = bits;
  const float g = __half2float(__half(r));
  g_gelu_tab[i] = __half_as_ushort(f32_to_f16_rn(mixdq_geluf(g)));
}

// The table is built once per device by the first launch that needs it.  `done[dev]` is only set once
// the table IS in memory for every stream: outside a capture the init kernel runs on the caller's
// stream and is waited for (one host synchronisation per device and process, at first use).  Inside a
// stream capture nothing may synchronise and the kernel is only RECORDED: the init is recorded in front
// of the FIRST consumer of that capture (idempotent: every replay rewrites the same bits; later
// consumers of the same capture are ordered behind it by the stream -- `in_capture[dev]` remembers the
// capture's id, so a UNet graph captured cold carries one init kernel, not seventy that rewrite the
// table under each other's readers) and `done` stays false -- an eager launch or another capture before
// the first replay builds the table itself.  mixdq_amd builds it eagerly (mixdq_gelu_table from
// SDXLUNet.prepare_fused_), so its captures record none.
inline int ensure_gelu_table(hipStream_t stream) {
  static bool done[64] = {};
  static unsigned long long in_capture[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
  if (done[dev]) return MIXDQ_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  unsigned long long cap_id = 0;
  if (hipStreamGetCaptureInfo(stream, &cap, &cap_id) != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap != hipStreamCaptureStatusNone && cap_id != 0 && in_capture[dev] == cap_id) return MIXDQ_OK;
  gelu_table_init_kernel<<<(2 * kGeluTabMag + 255) / 256, 256, 0, stream>>>();
  if (hipGetLastError() != hipSuccess) return MIXDQ_ERR_LAUNCH;
  if (cap == hipStreamCaptureStatusNone) {
    if (hipStreamSynchronize(stream) != hipSuccess) return MIXDQ_ERR_LAUNCH;
    done[dev] = true;
  } else {
    in_capture[dev] = cap_id;
  }
  return MIXDQ_OK;
}

// Two GEGLU outputs at a time from packed fp16 pairs (value xw, gate gw): the bytes q0 | q1 << 8 of
// quantize(f16(x * f16(gelu(g)))).  The epilogue is VALU-bound (with one workgroup per CU nothing runs
// beside it), so the element chain is kept short:
//   * HOW = 0, table in LDS, every |g| < 16 (decided per wave for a run of elements): the entry IS
//     f16(gelu(g));
//   * HOW = 1, table in LDS, some |g| >= 16 / inf / NaN in the run: g itself, or 0 * g for negative
//     gates (-0; NaN for -inf / NaN) -- the specification's values there -- selected on the bits,
//     branch-free (a NaN's payload does not matter: the product is NaN and quantizes to 0);
//   * HOW = 2, no table (tiles that share their CU): the arithmetic of include/mixdq_math.h;
//   * the product of two fp16 values is exact in FP32 (22 significant bits), so "FP32 multiply, round
//     to fp16" is v_pk_mul_f16 -- one instruction for the pair (fp16 denormals are on);
//   * the clamped integers are packed by v_perm_b32 (low byte of each), no masking.
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool geglu_any_far(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3) {
  auto mags = [](uint32_t w) { return __builtin_bit_cast(v2u16, w & 0x7fff7fffu); };
  const v2u16 m = __builtin_elementwise_max(__builtin_elementwise_max(mags(g0), mags(g1)),
                                            __builtin_elementwise_max(mags(g2), mags(g3)));
  return max((uint32_t)m[0], (uint32_t)m[1]) >= (uint32_t)kGeluTabMag;   // eight gates, any |g| >= 16
}
template <int HOW, bool UNFUSED>
__device__ __forceinline__ uint32_t geglu_pair(uint32_t xw, uint32_t gw, const char* Tb, float s_inv,
                                               float zpq) {
  v2h ge;
  if constexpr (HOW == 2) {
    const v2h gh = __builtin_bit_cast(v2h, gw);
    const v2f g2 = geluf2(v2f{(float)gh[0], (float)gh[1]});
    ge = v2h{(_Float16)f32_to_f16_rn(g2[0]), (_Float16)f32_to_f16_rn(g2[1])};
  } else {
    const uint32_t m0 = gw & 0x7fffu, n0 = (gw >> 15) & 1u, m1 = (gw >> 16) & 0x7fffu, n1 = gw >> 31;
    const uint32_t i0 = n0 * kGeluTabMag + (HOW == 0 ? m0 : min(m0, (uint32_t)kGeluTabMag - 1));
    const uint32_t i1 = n1 * kGeluTabMag + (HOW == 0 ? m1 : min(m1, (uint32_t)kGeluTabMag - 1));
    uint32_t t0 = *reinterpret_cast<const uint16_t*>(Tb + 2 * i0);
    uint32_t t1 = *reinterpret_cast<const uint16_t*>(Tb + 2 * i1);
    if constexpr (HOW == 1) {
      const uint32_t f0 = n0 ? (m0 >= 0x7c00u ? 0xfe00u : 0x8000u) : m0;
      const uint32_t f1 = n1 ? (m1 >= 0x7c00u ? 0xfe00u : 0x8000u) : m1;
      t0 = m0 >= (uint32_t)kGeluTabMag ? f0 : t0;
      t1 = m1 >= (uint32_t)kGeluTabMag ? f1 : t1;
    }
    ge = __builtin_bit_cast(v2h, t0 | (t1 << 16));
  }
  v2h y = __builtin_bit_cast(v2h, xw) * ge;
  asm("" : "+v"(y));
  // (clamp and packing of the pair in ONE instruction, v_ashr_pk_i8_i32: the pass is VALU-bound)
  const float x0 = (float)y[0], x1 = (float)y[1];
  const float t0 = UNFUSED ? __fadd_rn(__fmul_rn(x0, s_inv), zpq) : __builtin_fmaf(x0, s_inv, zpq);
  const float t1 = UNFUSED ? __fadd_rn(__fmul_rn(x1, s_inv), zpq) : __builtin_fmaf(x1, s_inv, zpq);
  return __builtin_amdgcn_ashr_pk_i8_i32((int)__builtin_rintf(t0), (int)__builtin_rintf(t1), 0);
}
// four outputs: the bytes of one dword
template <int HOW, bool UNFUSED>
__device__ __forceinline__ uint32_t geglu_quad(uint2 xq, uint2 gq, const char* Tb, float s_inv, float zpq) {
  const uint32_t lo = geglu_pair<HOW, UNFUSED>(xq.x, gq.x, Tb, s_inv, zpq);
  const uint32_t hi = geglu_pair<HOW, UNFUSED>(xq.y, gq.y, Tb, s_inv, zpq);
  return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// The GEGLU epilogue's LDS: the INT8 output tile (rows of BN / 2 bytes + 16) and, for the tiles that have
// their CU to themselves anyway (stages > 80 KB), the GELU table behind it -- both inside the K-tile
// stage buffers, which are free by then.
constexpr int geglu_tile_bytes(int BM, int BN) { return (BM * (BN / 2 + 16) + 1023) / 1024 * 1024; }
template <int BM, int BN, int BK, int STAGES>
constexpr bool igemm_gelu_table_fits() {
  return BN % 32 == 0 && STAGES * (BM + BN) * BK > 80 * 1024 &&
         geglu_tile_bytes(BM, BN) + kGeluTabBytes <= STAGES * (BM + BN) * BK;
}

template <int BM, int BN, int BK, int STAGES>
constexpr int igemm_main_bytes() {   // K-tile stages, overlaid by the epilogue's fp16 tile
  return (STAGES * (BM + BN) * BK > BM * (BN * 2 + 16)) ? STAGES * (BM + BN) * BK : BM * (BN * 2 + 16);
}
template <int BM, int BN, int BK, int STAGES>
constexpr int igemm_smem_bytes() {   // + the per-channel epilogue vectors: bias0, scale, bias
  return igemm_main_bytes<BM, BN, BK, STAGES>() + BN * 12;
}

// Waves per SIMD a tile configuration is meant to run at (second argument of __launch_bounds__): as
// many workgroups as its LDS lets a CU hold -- the co-resident workgroups are what hides one's epilogue
// under another's main loop -- unless the accumulators alone would not fit the registers that leaves.
// Without the bound the register allocator spends whatever makes the (straight-line) epilogue fastest:
// the 256x128 tile went from 104 to 170 VGPRs and lost its second workgroup per CU (batch 8:
// GEMM+GEGLU 177 -> 207 us, tools/stamp_report.py + same-box A/B).
// AQ: the FP16 activations of max(STAGES - 1, 2) K-tiles wait in registers (8 per 1-KiB piece of the INT8
// image and K-tile) and the quantizing pass needs ~24 more: the bound leaves room for them -- a spill of a
// register whose load is in flight would read garbage (tools/check_aq_isa.py).
template <int BM, int BN, int BK, int STAGES, int NWAVES, int ACC_REGS, bool AQ = false>
constexpr int igemm_waves_per_simd() {
  int wg = (160 * 1024) / (igemm_main_bytes<BM, BN, BK, STAGES>() + BN * 12);
  int w = wg * NWAVES / 4;
  if (w > 8) w = 8;
  if (w < 1) w = 1;
  const int aq_regs = AQ ? (STAGES - 1 >= 2 ? STAGES - 1 : 2) * (BM * BK / 1024 / NWAVES) * 8 + 24 : 0;
  while (w > 1 && ACC_REGS + 48 + aq_regs > 512 / w) --w;
  return w;
}

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a
// compile-time constant in every iteration (register arrays indexed by it stay in registers)
template <int N, int I = 0, class F>
__device__ __forceinline__ void igemm_unrolled(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    igemm_unrolled<N, I + 1>(f);
  }
}

// two packed fp16 + two packed fp16, each lane as torch's half add: f32 add, one rounding
__device__ __forceinline__ uint32_t add_f16x2(uint32_t a, uint32_t b) {
  const v2h ah = *reinterpret_cast<const v2h*>(&a), bh = *reinterpret_cast<const v2h*>(&b);
  v2f r = __builtin_convertvector(ah, v2f) + __builtin_convertvector(bh, v2f);
  asm("" : "+v"(r));
  const v2h h = __builtin_convertvector(r, v2h);
  return *reinterpret_cast<const uint32_t*>(&h);
}

// Kernel-argument preload (gfx950: the dispatcher can place the first 14 dwords of the argument block in
// scalar registers before the first wave starts; hipcc -mllvm -amdgpu-kernarg-preload-count=14 marks the
// leading SCALAR arguments -- a by-value struct is never preloaded).  The operands the first memory
// request of a launch depends on are therefore passed a second time, as leading scalars: the per-lane
// staging offsets and the prologue DMAs need nothing else, and the rest of the block (one scalar-cache
// round trip, ~0.5 us cold) arrives under them.
#ifndef MIXDQ_KP
#define MIXDQ_KP 1
#endif
#if MIXDQ_KP
#define MIXDQ_IGEMM_HEAD_PARAMS                                                                      \
  const int8_t* __restrict__ hA, const int8_t* __restrict__ hWt, int64_t hM, int hN, int hKtot,       \
      int htm_gm, int htn, const float* __restrict__ hb0, const float* __restrict__ hsc,
// (the super-row height of the tile map rides in the top byte of the preloaded tiles_m word)
#define MIXDQ_IGEMM_HEAD_TAKE(p)                                                                     \
  do { (p).A = hA; (p).Wt = hWt; (p).M = hM; (p).N = hN; (p).Ktot = hKtot;                            \
       (p).tiles_m = htm_gm & 0xffffff; (p).gm = (int)((unsigned)htm_gm >> 24);                        \
       (p).tiles_n = htn; (p).bias0 = hb0; (p).scale = hsc; } while (0)
#define MIXDQ_IGEMM_HEAD_ARGS(p)                                                                     \
  (p).A, (p).Wt, (p).M, (p).N, (p).Ktot, (int)((unsigned)(p).tiles_m | ((unsigned)(p).gm << 24)),     \
      (p).tiles_n, (p).bias0, (p).scale,
#else
#define MIXDQ_IGEMM_HEAD_PARAMS
#define MIXDQ_IGEMM_HEAD_TAKE(p) do {} while (0)
#define MIXDQ_IGEMM_HEAD_ARGS(p)
#endif

// FAST (Linear only): K % BK == 0 and every operand offset fits 32 bits.  Then the staging needs
// no per-K-tile vector arithmetic at all: each lane keeps one constant 32-bit byte offset per
// DMA piece and the K-tile advance is a scalar add on the uniform base pointer (saddr form of
// global_load_lds).  Rows past M / N are clamped to the last valid row instead of reading the
// zero page: their accumulators are never stored.
//
// W4: the weight operand is stored as packed signed 4-bit values (the
// reference has no W4 kernel: its 4-bit layers fall back to FP16).  Layout "nibble-planar per 8":
// byte j (0..3) of each 4-byte group holds  k[8g+j] in its HIGH nibble and k[8g+4+j] in its LOW
// nibble (two's complement).  The packed bytes go through the same LDS-DMA pipeline at half the
// bytes; a fragment read is one ds_read_b64 and the unpack is 3 VALU ops per packed dword:
//     hi = w & 0xF0F0F0F0          -> int8 values 16 * q[8g .. 8g+3]
//     lo = (w << 4) & 0xF0F0F0F0   -> int8 values 16 * q[8g+4 .. 8g+7]
// i.e. the MFMA runs on 16*q (still int8, |16 q| <= 128) and the epilogue uses bias0 * 16 and
// scale / 16 -- power-of-two factors, so every FP32 rounding is that of the unscaled arithmetic.
// KSPLIT = 2 (64x64 tiles only): two groups of WM x WN waves share the tile and split every K-tile's
// k-steps between them; group 1's accumulators are added to group 0's through LDS before the
// epilogue (int32: exact, order-free).  Twice the waves for the same tile halves the per-K-tile
// instruction chain each wave runs -- the thing that bounds the small GEMMs.
// MT: the MFMA shape -- 32 = v_mfma_i32_32x32x32_i8, 16 = v_mfma_i32_16x16x64_i8 (same int8 rate; wave
// tiles in multiples of 16, so block tiles such as 64x80 / 64x240 / 128x80 that put EXACTLY one
// workgroup on every CU for the UNet's N = 1280 / 3840 / 640 layers at M = 1024 / 4096).  These
// launches are bound by what one CU can pull from L2 into LDS (~70-100 GB/s), i.e. by
// (BM + BN) * K bytes per workgroup and the number of rounds; an exact-fit tile minimises both.
//
// F16: the same kernel on FP16 operands with FP32 accumulation (v_mfma_f32_32x32x16_f16 /
// 16x16x32): a k-step is the same 32 / 64 BYTES per row and a lane's fragment the same 16 bytes,
// so staging, swizzle and fragment reads are unchanged -- all sizes (K, C, BK) are in bytes.
// Used for the layers the reference leaves in FP16 (no activation quantizer: conv_in / conv_out,
// the act-protected ff.net.2 ..., nn/Linear.py:155-156): D = f16(acc + bias) [+ residual].
// LNQ: balanced tree over the values of the U (4, 8 or 16) lanes of a 16-lane row -- pairs at distance 1, 2, 4, 8,
// the order of the LayerNorm specification -- by DPP (quad permutes, then half-row and row mirrors: after two
// levels a quad holds one value, after three a half-row does, so a mirror reads the partner the xor would);
// every lane of the row returns the total.  (As __shfl_xor -- ds_bpermute, the LDS crossbar -- the two
// reductions of the row statistics took 1.9 us of the launch: tools/stamp_report.py --ln.)
template <int CTRL>
__device__ __forceinline__ float ln_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ln_row_tree(float t, int U) {
  t = __fadd_rn(t, ln_dpp<0xB1>(t));                 // quad_perm [1,0,3,2]: distance 1
  t = __fadd_rn(t, ln_dpp<0x4E>(t));                 // quad_perm [2,3,0,1]: distance 2
  if (U > 4) t = __fadd_rn(t, ln_dpp<0x141>(t));     // row_half_mirror: the other quad of the half-row
  if (U > 8) t = __fadd_rn(t, ln_dpp<0x140>(t));     // row_mirror: the other half-row
  return t;
}

// AQ: the counted wait for a slot's register loads, tied to the registers (see a_load in igemm_kernel)
template <int N, int SLOT>
__device__ __forceinline__ void aq_wait2(v4i& a, v4i& b) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%2) ; AQWAIT %3 %0 %1" : "+v"(a), "+v"(b) : "n"(N), "n"(SLOT) : "memory");
}
template <int N, int SLOT>
__device__ __forceinline__ void aq_wait4(v4i& a, v4i& b, v4i& c, v4i& d) {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%4) ; AQWAIT %5 %0 %1 %2 %3"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N), "n"(SLOT) : "memory");
}

template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool CONV, bool FAST, bool W4,
          int KSPLIT = 1, int MT = 32, bool F16 = false, bool ATT = false, bool PHASED = false,
          bool GROUPED = false, bool AQ = false, bool LNQ = false>
__global__ __launch_bounds__(
    64 * WM * WN * KSPLIT,
    (ATT ? 2 : igemm_waves_per_simd<BM, BN, BK, STAGES, WM * WN * KSPLIT,
                                    (BM / WM / MT) * (BN / WN / MT) * (MT == 32 ? 16 : 4), AQ>()))
void igemm_kernel(MIXDQ_IGEMM_HEAD_PARAMS const IgemmParams p_in) {
  static_assert(!PHASED || (BM == 256 && BN == 256 && BK == 128 && STAGES == 2 &&
                            WM == 2 && WN == 4 && KSPLIT == 1 && MT == 16 && FAST && !CONV && !W4 &&
                            !F16 && !ATT),
                "the phased loops are written for the 256x256 tile, 2 x 4 waves of 128x64");
  static_assert(MT == 32 || BK == 128 || PHASED, "the 16x16x64 fragment reads are laid out for 128-byte rows");
  static_assert(!ATT || (BM == 64 && BN == 128 && NWAVES_OF(WM, WN, KSPLIT) == 8 && MT == 32 && !CONV &&
                         !F16), "the attention epilogue is written for the 64x128 8-wave tile");
  static_assert(!(CONV && FAST), "the fast staging path is for Linear");
  // AQ -- quantize-in-prologue (replaces the reference's quantize launch in front of every layer,
  // nn/Linear.py:162-176): the activation operand is read as FP16 into registers (16 bytes = 8 values per
  // lane and load), quantized there -- q = sat8(rint(fma(x, s_inv, zp))), the arithmetic of
  // csrc/quantize.hip in both rounding variants -- and written to the SAME swizzled LDS image the
  // LDS-DMA of an INT8 operand produces (8 bytes per lane, lane-linear: conflict-free), one K-tile
  // ahead of its use; the weights stay on LDS-DMA.  Fragment reads, MFMAs and epilogue are unchanged.
  static_assert(!AQ || (FAST && !CONV && !F16 && !ATT && !PHASED && !GROUPED),
                "the quantizing activation stage is built for the Linear fast path");
  static_assert(!LNQ || (FAST && !CONV && !F16 && !ATT && !PHASED && !GROUPED && !AQ && !W4 && BN % 16 == 0),
                "the LayerNorm epilogue is built for the exact-fit Linear tiles");
  // every argument requested at once (common.h): 1.2-2.6 us from wave start to the first DMA before.
  // LATE_ARGS (Linear fast path with preloaded head arguments): the tile map, the per-lane staging offsets
  // and the prologue DMAs need the preloaded scalars only, so the rest of the argument block is asked
  // for (and waited for) BEHIND the prologue DMAs -- its scalar-cache round trip runs under them.
  constexpr bool LATE_ARGS = MIXDQ_KP && FAST && !ATT && !GROUPED && !AQ;   // (AQ: the row stride and row map of the FP16 operand are not among the preloaded arguments)
  auto args_now = [&]() {
#if MIXDQ_KP
    MIXDQ_ARGS_NOW(p_in.bias, p_in.table, p_in.zp, p_in.D, p_in.groups, p_in.Dq, p_in.res);
#else
    MIXDQ_ARGS_NOW(p_in.A, p_in.Wt, p_in.bias0, p_in.scale, p_in.bias, p_in.table, p_in.zp, p_in.D,
                   p_in.M, p_in.N, p_in.Ktot, p_in.tiles_m, p_in.tiles_n, p_in.groups, p_in.Dq, p_in.res);
#endif
    MIXDQ_ARGS_NOW(p_in.H, p_in.W, p_in.C, p_in.R, p_in.S, p_in.P, p_in.Q, p_in.stride, p_in.pad,
                   p_in.grp_rows, p_in.grp_stride, p_in.grp_off, p_in.res_div, p_in.unfused,
                   p_in.g_sinv, p_in.g_zp);
    if constexpr (AQ) MIXDQ_ARGS_NOW(p_in.a_sinv, p_in.a_zp);
    if constexpr (LNQ) {
      MIXDQ_ARGS_NOW(p_in.ln_gamma, p_in.ln_beta, p_in.ln_eps, p_in.ln_nq, p_in.ln_h, p_in.ln_part, p_in.ln_cnt,
                     p_in.ln_local);
      MIXDQ_ARGS_NOW(p_in.ln_sinv[0], p_in.ln_sinv[1], p_in.ln_sinv[2], p_in.ln_zp[0], p_in.ln_zp[1], p_in.ln_zp[2],
                     p_in.ln_q[0], p_in.ln_q[1], p_in.ln_q[2]);
    }
    if constexpr (ATT) {
      MIXDQ_ARGS_NOW(p_in.att_k, p_in.att_v, p_in.att_k_bs, p_in.att_v_bs, p_in.att_k_rs, p_in.att_v_rs,
                     p_in.att_tkv, p_in.att_tq, p_in.att_scale_log2, p_in.att_out, p_in.att_sinv,
                     p_in.att_zp);
    }
  };
  if constexpr (!LATE_ARGS) args_now();
#ifndef MIXDQ_KARG_WARM
#define MIXDQ_KARG_WARM 1
#endif
  // LATE_ARGS: the rest of the argument block is read by scalar loads that the compiler issues next to the
  // wait behind the first prologue stage -- cold, that is a trip to memory.  Four lanes touch its cache lines
  // with a vector load right here, so that the scalar loads find them in L2.
  int karg_warm = 0;
  if constexpr (LATE_ARGS && MIXDQ_KARG_WARM) {
    const auto ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    if ((threadIdx.x & ~3u) == 0)
      karg_warm = *(const __attribute__((address_space(4))) int*)(ka + 64 + 64 * (threadIdx.x & 3));
  }
  IgemmParams p = p_in;
  MIXDQ_IGEMM_HEAD_TAKE(p);
  // padded convs (table mode): the activation zero point, as a scalar load at entry (it was a dependent trip
  // to memory in front of the accumulator pass)
  float zpv_early = 0.f;
  if constexpr (CONV) {
    if (p_in.table != nullptr) zpv_early = *(const __attribute__((address_space(4))) float*)p_in.zp;
  }
  // ATT: the output quantizer's scalars, as scalar loads at entry -- read behind the attention epilogue they
  // were a dependent trip to memory at the very end of every to_q + cross-attention launch
  float att_sinv_early = 0.f, att_zp_early = 0.f;
  if constexpr (ATT) {
    if (p_in.att_sinv != nullptr) {
      att_sinv_early = *(const __attribute__((address_space(4))) float*)p_in.att_sinv;
      att_zp_early = *(const __attribute__((address_space(4))) float*)p_in.att_zp;
    }
  }
  float aq_sinv = 0.f, aq_zp = 0.f;
  bool aq_unfused = false;
  if constexpr (AQ) {
    // (through the constant address space: scalar loads -- as vector loads they would sit in front of the
    //  prologue requests and the compiler's wait for them would drain every one of those)
    aq_sinv = *(const __attribute__((address_space(4))) float*)p_in.a_sinv;
    aq_zp = *(const __attribute__((address_space(4))) float*)p_in.a_zp;
    aq_unfused = p_in.unfused != 0;
  }
  int nwg = p.tiles_m * p.tiles_n;   // == gridDim.x (which would be one more dependent load)
  if constexpr (GROUPED) {          // one member of a grouped launch (wave-uniform scalar loads)
    const mixdq_gemm_group g = p_in.groups[blockIdx.y];
    p.Wt = g.W; p.bias0 = g.bias0; p.scale = g.scale; p.bias = (const __half*)g.bias_f16_or_null;
    p.D = (__half*)g.D_f16; p.N = g.N;
    p.tiles_n = (g.N + BN - 1) / BN;
    nwg = p.tiles_m * p.tiles_n;
    if ((int)blockIdx.x >= nwg) return;   // the grid is sized for the widest member
  }
  static_assert(!(F16 && W4), "packed weights are an INT8-path format");
  static_assert(MT == 32 || MT == 16, "MFMA shapes: 32x32x32 or 16x16x64");
  constexpr int NWAVES = WM * WN * KSPLIT, NTHREADS = 64 * NWAVES;
  // registers per lane the launch bound leaves this kernel (512 per SIMD lane over the waves per SIMD)
  constexpr int REG_CAP = 512 / (ATT ? 2 : igemm_waves_per_simd<BM, BN, BK, STAGES, WM * WN * KSPLIT,
                                                                 (BM / WM / MT) * (BN / WN / MT) * (MT == 32 ? 16 : 4), AQ>());
  constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile (WM x WN waves)
  constexpr int TM = WTM / MT, TN = WTN / MT;     // MT x MT MFMA tiles per wave
  constexpr int KSTEP = MT == 32 ? 32 : 64;       // k-values one MFMA consumes
  constexpr int CPS = KSTEP / 16;                 // 16-byte fragment chunks per k-step
  constexpr int ACC = MT == 32 ? 16 : 4;          // accumulator registers per MFMA tile
  constexpr int WB = W4 ? 2 : 1;                  // weights per stored byte
  constexpr int A_STAGE = BM * BK, B_STAGE = BN * BK / WB, STAGE = A_STAGE + B_STAGE;
  // LDS-DMA pieces (1 KiB = one wave-instruction).  Activations: the same count on every wave.
  // Weights: piece q goes to wave q % NWAVES, so any piece count works (BN = 80, 240, 320; packed
  // W4 stages of half the bytes); waves below PB % NWAVES issue one more and wait for one more.
  constexpr int A_NI = A_STAGE / 1024 / NWAVES;
  constexpr int PB = B_STAGE / 1024;
  constexpr int B_LO = PB / NWAVES, B_REM = PB % NWAVES, B_NI = B_LO + (B_REM ? 1 : 0);
  constexpr int CS_STRIDE = BN * 2 + 16;          // epilogue tile row stride (bytes)
  // LNQ: behind the fp16 tile -- [BM][2] row statistics (mean, rstd), then [BM][BN / 16][2] group statistics
  constexpr int LN_GRP_OFF = BM * CS_STRIDE + BM * 8;
  static_assert(!LNQ || LN_GRP_OFF + BM * (BN / 16) * 8 <= igemm_main_bytes<BM, BN, BK, STAGES>(),
                "the LayerNorm tables fit behind the tile");
  constexpr int PRE = STAGES - 1;                 // K-tiles in flight ahead of the one computed
  static_assert(A_NI >= 1 && A_NI * 1024 * NWAVES == A_STAGE && PB >= 1 && PB * 1024 == B_STAGE,
                "whole 1-KiB DMA pieces; the activation pieces divide evenly over the waves");
  static_assert(TM >= 1 && TN >= 1 && TM * MT * WM == BM && TN * MT * WN == BN,
                "wave tiles are whole MFMA tiles");
  // AQ: a K-tile's activations are 2 * A_NI register loads per lane (16 bytes of FP16 -> 8 bytes of the INT8
  // image each) instead of A_NI DMA pieces, requested AD K-tiles ahead and parked in registers.
  constexpr int A_OPS = AQ ? 2 * A_NI : A_NI;     // vector-memory instructions per wave and K-tile, activations
  constexpr int AD = PRE >= 2 ? PRE : 2;          // AQ: K-tiles of raw FP16 activations in flight
  static_assert(STAGES >= 2 && (PRE - 1) * (A_OPS + B_NI) <= 63, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // igemm_smem_bytes<...>()

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wid / (WM * WN);                 // k-split group (0 when KSPLIT == 1)
  const int wm = (wid % (WM * WN)) / WN, wn = wid % WN;
  MIXDQ_STAMP_AT(0);

  // ---- XCD- and L2-aware tile map.  Blocks are dealt round-robin over the 8 XCDs (bid % 8), each
  //      with its own 4 MiB L2: give every XCD a contiguous run of the tile sequence (bijective
  //      remap), and order that sequence in super-rows of GM m-tiles (m fastest inside, then n),
  //      so the ~100 blocks an XCD runs at once cover a near-square patch of the output and share
  //      both their activation rows and their weight panels in that L2 instead of streaming all
  //      of A for every column of tiles.
  const int bid = blockIdx.x;
  const int xcd = bid % kNumXCD, q8 = nwg / kNumXCD, r8 = nwg % kNumXCD;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kNumXCD;
  const int GM = p.gm;
  const int per_group = GM * p.tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GM;
  const int gsz = min(GM, p.tiles_m - first_m);
  const int rem = wg - group * per_group;
  int tile_n = rem / gsz, tile_m = first_m + (rem - tile_n * gsz);
  if constexpr (LNQ) {
    // the column tiles of a row block exchange their LayerNorm records: n fastest, so that an XCD's contiguous
    // run of the sequence holds WHOLE row blocks and the records can meet in that XCD's L2.  (Short K only --
    // ln_local, the launcher's choice: an XCD then streams ALL of W, which cost the K = 5120 launch 3.5 us.)
    if (p.ln_local) {
      tile_m = wg / p.tiles_n;
      tile_n = wg - tile_m * p.tiles_n;
    }
  }
  const int64_t m0 = (int64_t)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- residual operand of the epilogue, requested up front on the small tiles: it comes cold (another
  //      kernel just wrote it), and read inside the store loop its ~1 us of latency sat at the very
  //      end of every to_out / ff.net.2 / conv2 launch of the batch-1 chain.  Requested in front of the
  //      prologue DMAs (or, LATE_ARGS, right behind them: then the first counted wait also waits for
  //      these few loads, which were issued next to the stage it waits for) and used behind the main
  //      loop's final vmcnt(0).
  // lane -> (tile row, 16-byte k-chunk of the k-step) of an MFMA fragment: 32x32x32 has the row on lane & 31
  // and two chunks (lane >> 5), 16x16x64 the row on lane & 15 and four chunks (lane >> 4); in the
  // accumulator the same lane holds output row `lrow` and 4 consecutive channels per register quad
  const int lrow = MT == 32 ? (lane & 31) : (lane & 15);
  const int lkq = MT == 32 ? (lane >> 5) : (lane >> 4);
  constexpr int RES_ITERS = (BM * (BN / 8) + 64 * WM * WN * KSPLIT - 1) / (64 * WM * WN * KSPLIT);
  constexpr bool RES_PRE = !ATT && !GROUPED && RES_ITERS <= 2;
  uint4 res_pre[RES_PRE ? RES_ITERS : 1];
  bool res_pre_on = false;
  // ---- this thread's slice of the per-channel epilogue vectors (threads < BN / 4), requested up front
  //      too and parked in registers: they go to LDS behind the main loop.  (Stored to LDS right
  //      here, as round 2 did, the store's vmcnt(0) made the first waves wait for EVERY prologue
  //      stage -- five K-tiles on the six-stage tile -- before the first K-tile could be computed.)
  //      P_B0: bias0[n] (table mode: the full-window class row), P_SC: scale[n], P_BS: bias[n].
  bool has_bias = false, use_table = false;
  int full_cls = 0;
  int ln_epoch = 0;        // LNQ: the workspace's launch counter (this launch's records carry ln_epoch + 1)
  v4f pre_b0, pre_sc;
  uint2 pre_bs;
  const bool pre_on = tid < BN / 4;
  const bool pre_in = n0 + tid * 4 < p.N;
  auto early_loads = [&]() {
    if constexpr (LNQ) ln_epoch = __hip_atomic_load(p.ln_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    has_bias = p.bias != nullptr;
    use_table = p.table != nullptr;
    full_cls = (((p.R - 1)) * p.S) * p.S + (p.S - 1);   // rlo=0, rhi=R-1, slo=0, shi=S-1
    res_pre_on = RES_PRE && p.res != nullptr && (p.N & 7) == 0;
    if constexpr (RES_PRE) {
      if (res_pre_on) {   // wave-uniform; addresses clamped into the tensor instead of predicated, so
#pragma unroll            // that no select sits between the load and its use after the main loop
        for (int it = 0; it < RES_ITERS; ++it) {
          const int idx = min(tid + it * (64 * WM * WN * KSPLIT), BM * (BN / 8) - 1);
          const int row = idx / (BN / 8), cc = idx - row * (BN / 8);
          const int64_t m = min(m0 + row, p.M - 1);
          const int n = n0 + cc * 8 < p.N ? n0 + cc * 8 : 0;
          res_pre[it] = *reinterpret_cast<const uint4*>(
              p.res + (p.res_div == 1 ? m : m / p.res_div) * p.N + n);
        }
      }
    }
    if (pre_on) {
      const int n = pre_in ? n0 + tid * 4 : 0;
      if constexpr (!F16) {
        const float* b0src = p.bias0;
        if constexpr (CONV) if (use_table) b0src = p.table + (int64_t)full_cls * p.N;
        pre_b0 = *reinterpret_cast<const v4f*>(b0src + n);
        pre_sc = *reinterpret_cast<const v4f*>(p.scale + n);
      }
      if (has_bias) pre_bs = *reinterpret_cast<const uint2*>(p.bias + n);
    }
  };
  if constexpr (!LATE_ARGS) early_loads();

  const char* zero = reinterpret_cast<const char*>(&g_zero16);
  const int Ktot = p.Ktot;
  // (Round 4, measured and removed -- "K rotation": every tile of the Linear fast path started at its own K-tile,
  //  (m + 3 n) mod nk, and wrapped, so that the workgroups that share an operand panel in an XCD's L2 would not
  //  all request a line while its first request is still on its way to memory.  Integer accumulation is exact in
  //  any order, the results were bit-identical -- and no launch got faster: (8192, 1280, 5120) 59.2 -> 63.2 us,
  //  the rest within +-1 %, batch-1 step 11.61 -> 11.69 ms.  The per-CU request rate, not the latency of a
  //  shared miss, is the bound.  profiles/r04_k_rotation_ab.txt)

  // ---- per-lane staging state -------------------------------------------------------------
  const int8_t* a_base[A_NI];   // linear: row pointer + chunk offset; conv: image base
  int a_k[A_NI];                // linear: chunk's k offset within the K-tile
  int a_h0[A_NI], a_w0[A_NI];   // conv: top-left input coordinate of the row's window
  int a_r[A_NI], a_s[A_NI], a_c[A_NI];   // conv: current tap and channel of this lane's chunk
  bool a_ok[A_NI];
#pragma unroll
  for (int j = 0; j < A_NI; ++j) {
    const int byte = (wid * A_NI + j) * 1024 + lane * 16;
    const int row = byte / BK;
    const int lc = ((byte % BK) >> 4) ^ swz<BK>(row);
    const int64_t m = m0 + row;
    a_ok[j] = m < p.M;
    if constexpr (!CONV) {
      a_base[j] = p.A + m * Ktot + lc * 16;
      a_k[j] = lc * 16;
      a_h0[j] = a_w0[j] = a_r[j] = a_s[j] = a_c[j] = 0;
    } else {
      const int pq = p.P * p.Q;
      const int64_t img = m / pq;
      const int rem = (int)(m - img * pq);
      const int pp = rem / p.Q, qq = rem - pp * p.Q;
      a_base[j] = p.A + img * ((int64_t)p.H * p.W * p.C);
      a_h0[j] = a_ok[j] ? pp * p.stride - p.pad : -(1 << 28);
      a_w0[j] = qq * p.stride - p.pad;
      const int kc = lc * 16;
      const int tap = kc / p.C;
      a_c[j] = kc - tap * p.C;
      a_r[j] = tap / p.S;
      a_s[j] = tap - a_r[j] * p.S;
      a_k[j] = 0;
    }
  }
  const int8_t* b_base[B_NI];
  int b_k[B_NI];
  bool b_ok[B_NI];
  uint32_t a_off32[A_NI], b_off32[B_NI];   // FAST: constant per-lane byte offsets
#pragma unroll
  for (int j = 0; j < B_NI; ++j) {
    // piece wid + NWAVES * j of the weight stage (the last j may fall past the stage on some
    // waves: its state is computed on clamped rows and never used)
    const int byte = (wid + NWAVES * j) * 1024 + lane * 16;
    int row, koff, boff;          // tile row; first k of this lane's 16-byte piece; its byte offset
    if constexpr (!W4) {
      row = byte / BK;
      const int lc = ((byte % BK) >> 4) ^ swz<BK>(row);
      koff = lc * 16;
      boff = lc * 16;
    } else {                      // packed: a 16-byte piece = 32 k-values = one MFMA k-step of a row
      constexpr int KSP = BK / 32;
      const int piece = byte >> 4;
      row = piece / KSP;
      const int ks = (piece % KSP) ^ ((row >> 3) & (KSP - 1));
      koff = ks * 32;
      boff = ks * 16;
    }
    const int n = n0 + row;
    b_ok[j] = n < p.N;
    b_base[j] = p.Wt + (int64_t)n * (Ktot / WB) + boff;
    b_k[j] = koff;
    b_off32[j] = (uint32_t)min(n, p.N - 1) * (uint32_t)(Ktot / WB) + boff;
  }
#pragma unroll
  for (int j = 0; j < A_NI; ++j) {
    const int byte = (wid * A_NI + j) * 1024 + lane * 16;
    const int row = byte / BK;
    const int lc = ((byte % BK) >> 4) ^ swz<BK>(row);
    const int64_t m = m0 + row;
    a_off32[j] = (uint32_t)(m < p.M ? m : p.M - 1) * (uint32_t)Ktot + lc * 16;
  }

  auto stage = [&](int buf, int kk) {
    char* As = smem + buf * STAGE;
    char* Bs = As + A_STAGE;
    if constexpr (FAST) {
      // tiles past the end of K (staged only to keep the vmcnt count uniform) re-read tile 0
      const int kk_u = __builtin_amdgcn_readfirstlane(kk < Ktot ? kk : 0);
      const int8_t* a_u = p.A + kk_u;
      const int8_t* b_u = p.Wt + kk_u / WB;
      if constexpr (!AQ) {
#pragma unroll
        for (int j = 0; j < A_NI; ++j) glds16(a_u + a_off32[j], As + (wid * A_NI + j) * 1024);
      }
#pragma unroll
      for (int j = 0; j < B_NI; ++j)
        if (j < B_LO || wid < B_REM) glds16(b_u + b_off32[j], Bs + (wid + NWAVES * j) * 1024);
      return;
    }
#pragma unroll
    for (int j = 0; j < A_NI; ++j) {
      const void* src;
      if constexpr (!CONV) {
        const bool ok = a_ok[j] && (kk + a_k[j] < Ktot);
        src = ok ? (const void*)(a_base[j] + kk) : (const void*)zero;
      } else {
        const int hh = a_h0[j] + a_r[j], ww = a_w0[j] + a_s[j];
        const bool ok = (unsigned)hh < (unsigned)p.H && (unsigned)ww < (unsigned)p.W &&
                        a_r[j] < p.R;
        src = ok ? (const void*)(a_base[j] + ((int64_t)(hh * p.W + ww) * p.C + a_c[j]))
                 : (const void*)zero;
        a_c[j] += BK;
        while (a_c[j] >= p.C) {
          a_c[j] -= p.C;
          if (++a_s[j] == p.S) { a_s[j] = 0; ++a_r[j]; }
        }
      }
      glds16(src, As + (wid * A_NI + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < B_NI; ++j) {
      if (j >= B_LO && wid >= B_REM) continue;   // wave-uniform: this wave has no such piece
      const bool ok = b_ok[j] && (kk + b_k[j] < Ktot);
      const void* src = ok ? (const void*)(b_base[j] + kk / WB) : (const void*)zero;
      glds16(src, Bs + (wid + NWAVES * j) * 1024);
    }
  };
  // counted wait for this wave's pieces of the oldest K-tile in flight, then the block barrier
  auto wait_tile = [&]() {
#if MIXDQ_ABLATE == 3
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    return;
#endif
    // (AQ: the INT8 image of this K-tile's activations was written by ordinary LDS stores one iteration
    //  earlier -- lgkmcnt(0) in front of the barrier makes every wave's stores visible behind it.  The
    //  weight pieces of K-tile kt are the LAST requests of iteration kt - PRE, behind that iteration's
    //  register loads, so the count is the same formula on A_OPS requests per K-tile.)
    if constexpr (AQ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (B_REM == 0) {
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PRE - 1) * (A_OPS + B_LO)) : "memory");
    } else {
      if (wid < B_REM)
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PRE - 1) * (A_OPS + B_NI)) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PRE - 1) * (A_OPS + B_LO)) : "memory");
    }
  };
  // ---- AQ: the activation stage in registers ---------------------------------------------------
  // Half i of piece j of this wave covers LDS bytes [(wid * A_NI + j) * 1024 + i * 512 + lane * 8, + 8) of the
  // stage's activation image: tile row `off / BK`, 8-byte slot `(off % BK) >> 3` of the row, i.e. half
  // `slot & 1` of the (swizzled) 16-byte chunk `slot >> 1` -- the 16 lanes of a row read one contiguous
  // 2 * BK-byte run of FP16 (whole 128-byte lines), the wave writes 512 contiguous bytes of LDS.
  uint32_t aq_off[AQ ? A_NI : 1][2];
  v4i araw[AQ ? STAGES : 1][AQ ? 2 * A_NI : 1];   // slot t % STAGES holds K-tile t (AD of them live at a time)
  if constexpr (AQ) {
#pragma unroll
    for (int j = 0; j < A_NI; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = (wid * A_NI + j) * 1024 + i * 512 + lane * 8;
        const int row = off / BK, slot = (off % BK) >> 3;
        const int k = (((slot >> 1) ^ swz<BK>(row)) << 4) + ((slot & 1) << 3);
        int64_t m = m0 + row < p.M ? m0 + row : p.M - 1;
        if (p.a_rowmap) {
          const int64_t gq = m / p.grp_rows;
          m = gq * p.grp_stride + p.grp_off + (m - gq * p.grp_rows);
        }
        aq_off[j][i] = (uint32_t)((m * p.a_ld + k) * 2);
      }
  }
  auto a_load = [&](auto slot_c, int kk) {          // K-tile at k offset kk -> registers (slot: compile time)
    if constexpr (AQ) {
      constexpr int SLOT = decltype(slot_c)::value;
      const int kk_u = __builtin_amdgcn_readfirstlane(kk < Ktot ? kk : 0);
      const char* a_u = reinterpret_cast<const char*>(p.A) + 2 * (int64_t)kk_u;
      // Inline asm, and the registers stay invisible to the compiler until a_wait() hands them over: with
      // ordinary loads hipcc's own wait-count pass loses the count across the unrolled K loop and puts
      // vmcnt(0) at the top of every K-tile -- the whole prefetch pipeline (weights included) serialised.
      // The price: between this request and a_wait() the compiler believes the registers already hold
      // their values, so it must neither copy nor spill them there; tools/check_aq_isa.py verifies on
      // the generated ISA that every register a load writes is next mentioned by its own wait.
#pragma unroll
      for (int j = 0; j < A_NI; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          v4i& dst = araw[SLOT][2 * j + i];       // (named first: clang does not capture a variable that
          const uint32_t off = aq_off[j][i];      //  a generic lambda uses in asm operands only)
          asm volatile("global_load_dwordx4 %0, %1, %2 ; AQLOAD %3"
                       : "=v"(dst) : "v"(off), "s"(a_u), "n"(SLOT) : "memory");
        }
    }
  };
  // counted wait for the register loads of slot SLOT (issued AD - 1 iterations ago, in front of that
  // iteration's weight DMAs), tied to the registers: every later use reads the waited-for value
  auto a_wait = [&](auto slot_c, auto first_c) {
    if constexpr (AQ) {
      constexpr int SLOT = decltype(slot_c)::value;
      constexpr bool FIRST = decltype(first_c)::value;     // the wait in front of the loop (PRE == 1: fewer requests behind it)
      static_assert(A_NI == 1 || A_NI == 2, "a_wait ties 2 or 4 register quads");
      // requests behind the awaited loads: that iteration's weight pieces (PRE == AD; PRE == 1 has them one
      // iteration later) and AD - 1 whole iterations.  ONE statement for every wave, with the count of the
      // waves that issue B_LO weight pieces (the waves with one more wait for a few requests more than
      // they need to): tied waits on two sides of a branch make the compiler copy the registers into
      // the branch -- in front of the wait, while the loads are in flight.
      constexpr int N = PRE >= 2 ? B_LO + (AD - 1) * (A_OPS + B_LO) : (FIRST ? A_OPS + B_LO : A_OPS + 2 * B_LO);
      v4i* r = araw[SLOT];
      if constexpr (A_NI == 1) aq_wait2<N, SLOT>(r[0], r[1]);
      else aq_wait4<N, SLOT>(r[0], r[1], r[2], r[3]);
    }
  };
  // (the quantizer's scalars: read at kernel entry, in front of every statement that clobbers memory, so
  //  that they are scalar loads -- as vector loads behind the prologue requests their wait would be vmcnt(0))
  auto a_convert = [&](auto slot_c, int buf) {      // registers -> INT8 image of LDS buffer `buf`
    if constexpr (AQ) {
      constexpr int SLOT = decltype(slot_c)::value;
      char* As = smem + buf * STAGE;
      auto q8 = [&](auto unf_c, const v4i& raw) -> uint2 {
        constexpr bool UNF = decltype(unf_c)::value;
        const v8h h = __builtin_bit_cast(v8h, raw);
        int q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (float)h[e];
          const float t = UNF ? __fadd_rn(__fmul_rn(x, aq_sinv), aq_zp) : __builtin_fmaf(x, aq_sinv, aq_zp);
          q[e] = (int)__builtin_rintf(t);           // v_rndne_f32 + v_cvt_i32_f32 (saturating, NaN -> 0)
        }
        // v_ashr_pk_i8_i32: two INT32 -> two saturated INT8 in one instruction (= common.h's clamp)
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __builtin_amdgcn_ashr_pk_i8_i32(q[2 * e], q[2 * e + 1], 0);
        return make_uint2(__builtin_amdgcn_perm(w[1], w[0], 0x05040100u),
                          __builtin_amdgcn_perm(w[3], w[2], 0x05040100u));
      };
      uint2 ov[2 * A_NI];
      if (aq_unfused) {                  // wave-uniform
#pragma unroll
        for (int x = 0; x < 2 * A_NI; ++x) ov[x] = q8(std::true_type{}, araw[SLOT][x]);
      } else {
#pragma unroll
        for (int x = 0; x < 2 * A_NI; ++x) ov[x] = q8(std::false_type{}, araw[SLOT][x]);
      }
#pragma unroll
      for (int j = 0; j < A_NI; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint2 o = ov[2 * j + i];
          // (inline asm: hipcc orders a compiler-generated LDS store behind EVERY LDS-DMA in flight --
          //  vmcnt(0), the weight prefetch serialised -- because it cannot tell that the store and the DMA
          //  pieces never overlap.  A 64-bit LDS store reads its data at issue: no hazard padding needed;
          //  wait_tile's lgkmcnt(0) covers it.)
          const uint32_t lds_a = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(
              As + (wid * A_NI + j) * 1024 + i * 512 + lane * 8);
          asm volatile("ds_write_b64 %0, %1" ::"v"(lds_a), "v"(o) : "memory");
        }
      asm volatile("; AQDONE %0" ::"n"(SLOT));
    }
  };

  using acc_i = typename std::conditional<MT == 32, v16i, v4i>::type;
  using acc_f = typename std::conditional<MT == 32, v16f, v4f>::type;
  using acc_t = typename std::conditional<F16, acc_f, acc_i>::type;
  auto mfma = [](const v4i& w, const v4i& x, acc_t c) -> acc_t {
    if constexpr (F16) {
      const v8h wh = __builtin_bit_cast(v8h, w), xh = __builtin_bit_cast(v8h, x);
      if constexpr (MT == 32) return __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, c, 0, 0, 0);
    } else {
      if constexpr (MT == 32) return __builtin_amdgcn_mfma_i32_32x32x32_i8(w, x, c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_i32_16x16x64_i8(w, x, c, 0, 0, 0);
    }
  };
  auto load_w = [&](const char* S0, int off) -> v4i {
    if constexpr (!W4) {
      return *reinterpret_cast<const v4i*>(S0 + off);
    } else {
      const uint2 w = *reinterpret_cast<const uint2*>(S0 + off);
      v4i r;
      r[0] = (int)(w.x & 0xF0F0F0F0u);
      r[1] = (int)((w.x << 4) & 0xF0F0F0F0u);
      r[2] = (int)(w.y & 0xF0F0F0F0u);
      r[3] = (int)((w.y << 4) & 0xF0F0F0F0u);
      return r;
    }
  };
  acc_t acc[TN][TM];

  // ---- main loop: STAGES LDS buffers, STAGES-1 K-tiles of LDS-DMA in flight.  Per K-tile ONE
  //      counted wait (never vmcnt(0)) + ONE raw s_barrier: the wait retires this wave's DMA
  //      pieces of tile kt, the barrier makes every wave's pieces visible and guarantees that
  //      buffer (kt-1) % STAGES is no longer read, so tile kt+PRE may be staged into it.
  //      Tiles past the end of K stage zero-page reads so the count stays uniform.
  const int nk = (Ktot + BK - 1) / BK;
  // ATT: the keys / values of this tile's two heads (two 64-key tiles each, K image then V image,
  // the layout of csrc/attention.hip) are requested first -- 8 DMA pieces per wave, older than
  // every K-tile piece, so the counted waits below retire them without further bookkeeping --
  // into LDS behind the stage buffers and the epilogue vectors.
  constexpr int ATT_OFF = ((igemm_smem_bytes<BM, BN, BK, STAGES>() + 1023) / 1024) * 1024;
  if constexpr (ATT) {
    const int hl = wid >> 2, part = wid & 3;            // head of the pair; (tile, K | V)
    const int t = part >> 1;
    const bool is_v = part & 1;
    const int64_t img = m0 / p.att_tq;                   // a tile never straddles two images
    const int rs = is_v ? p.att_v_rs : p.att_k_rs;
    const __half* base = (is_v ? p.att_v + img * p.att_v_bs : p.att_k + img * p.att_k_bs) +
                         (n0 + hl * kHeadDim);
    const int srow = lane >> 3, spos = lane & 7;
    char* dst = smem + ATT_OFF + hl * (2 * kStageBytes) + t * kStageBytes + (is_v ? kTileBytes : 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int krow = i * 8 + srow;                     // key row within the tile
      const int sw = is_v ? ((krow >> 1) & 1) << 2 : (krow >> 1) & 7;
      const int key = min(t * kKeys + krow, p.att_tkv - 1);   // absent keys: finite, masked below
      glds16(base + ((int64_t)key * rs + (spos ^ sw) * 8), dst + i * 1024);
    }
  }
  // ---- PHASED: the 256x256x128 tile in four phases per K-tile ---------------------------------
  // Each wave owns 128 x 64 of the output as 2 x 2 quadrants of 64 x 32 (16 MFMAs 16x16x64 over the
  // K-tile each); a phase = one quadrant: the LDS reads of the fragments that changed (snake order:
  // 12, 4, 8, 4 ds_read_b128), the DMA of one quarter of the NEXT K-tile, then the 16 MFMAs at raised
  // priority between two barriers.  Waves 4..7 run one barrier behind waves 0..3, so on every SIMD
  // one wave multiplies while the other reads and stages.  The next K-tile is staged in the four
  // "units" the phases consume -- activation rows of row-half mh of every wave, weight rows of
  // column-half nh -- in the order they are first read (A0, B0, B1, A1), two 1-KiB pieces per wave
  // each; three phases of DMA lead, counted waits (vmcnt(4): the two youngest units may fly).
  unsigned ph_a[2][2], ph_b[2][2];      // per-lane source offsets of (unit, piece)
  if constexpr (PHASED) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = wid * 2 + j;                                   // piece of the unit, 0..15
        const int ra = (q >> 3) * 128 + h * 64 + (q & 7) * 8 + (lane >> 3);
        const int rb = (q >> 2) * 64 + h * 32 + (q & 3) * 8 + (lane >> 3);
        const int64_t m = m0 + ra;
        const int n = n0 + rb;
        ph_a[h][j] = (uint32_t)(m < p.M ? m : p.M - 1) * (uint32_t)Ktot + (((lane & 7) ^ swz<BK>(ra)) << 4);
        ph_b[h][j] = (uint32_t)min(n, p.N - 1) * (uint32_t)Ktot + (((lane & 7) ^ swz<BK>(rb)) << 4);
      }
  }
  auto stage_unit = [&](int buf, int kk, int unit) {   // unit: 0 A0, 1 B0, 2 B1, 3 A1 (compile-time)
    const int kk_u = __builtin_amdgcn_readfirstlane(kk < Ktot ? kk : 0);
    char* S = smem + buf * STAGE;
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
  // LATE_ARGS: the rest of the argument block is waited for behind the FIRST prologue stage (the
  // address unit is busy with that stage's requests for ~0.5 us anyway), and the epilogue operands are
  // requested there: older than every later stage, so the first counted wait covers them with the
  // stage they were issued next to.
  auto late_args = [&]() {
    if constexpr (LATE_ARGS) {
      args_now();
      early_loads();
    }
  };
  // (`slot` is a constant once the K loop is unrolled by STAGES: the chain below folds to one call)
  auto aq_slot = [&](int slot, auto&& f) {
    if constexpr (AQ)
      igemm_unrolled<STAGES>([&](auto c) { if (decltype(c)::value == slot) f(c); });
  };
  if constexpr (AQ) {
    // iterations -AD .. -1 of the main loop's request pattern: activations of K-tile i + AD into registers,
    // then the weight DMA of K-tile i + PRE (PRE == AD, or PRE == 1 < AD == 2: no weights at i == -2)
    igemm_unrolled<AD>([&](auto t_c) {
      constexpr int T = decltype(t_c)::value;
      a_load(std::integral_constant<int, T % STAGES>{}, T * BK);
      if constexpr (T + PRE - AD >= 0) stage((T + PRE - AD) % STAGES, (T + PRE - AD) * BK);
    });
    a_wait(std::integral_constant<int, 0>{}, std::true_type{});
    a_convert(std::integral_constant<int, 0>{}, 0);     // K-tile 0's image
  } else if constexpr (!PHASED) {
    stage(0, 0);
    MIXDQ_STAMP_AT(12);
    late_args();
    MIXDQ_STAMP_AT(13);
#pragma unroll
    for (int s = 1; s < PRE; ++s) stage(s, s * BK);
  } else {
    stage_unit(0, 0, 0);
    stage_unit(0, 0, 1);
    MIXDQ_STAMP_AT(12);
    late_args();
    MIXDQ_STAMP_AT(13);
    stage_unit(0, 0, 2);
    stage_unit(0, 0, 3);
  }
  MIXDQ_STAMP_AT(1);
  // (everything below is needed by the main loop only: it is computed behind the prologue DMAs, whose
  //  requests are on their way meanwhile)
  // ---- fragment read offsets (loop invariant; the LDS buffer base is a compile-time constant
  //      after the K loop is unrolled by STAGES, so each ds_read_b128 needs no address arithmetic)
  // lane -> (tile row, 16-byte k-chunk of the k-step): 32x32x32 has the row on lane & 31 and two
  // chunks (lane >> 5), 16x16x64 the row on lane & 15 and four chunks (lane >> 4)
  constexpr int KS = BK / KSTEP / KSPLIT;            // k-steps of a K-tile this wave computes
  static_assert((BK / KSTEP) % KSPLIT == 0 && KS >= 1, "k-split groups take whole k-steps");
  int a_rd[TM][KS], b_rd[TN][KS];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * WTM + t * MT + lrow;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int ks = kg * KS + i;
      a_rd[t][i] = row * BK + (((ks * CPS + lkq) ^ swz<BK>(row)) << 4);
    }
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * WTN + t * MT + lrow;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int ks = kg * KS + i;
      const int c = ks * CPS + lkq;   // this lane's 16-k chunk of the K-tile
      if constexpr (!W4) {
        b_rd[t][i] = A_STAGE + row * BK + ((c ^ swz<BK>(row)) << 4);
      } else {   // packed: 16 k-values = 8 bytes, half (c & 1) of the 32-k piece c >> 1
        constexpr int KSP = BK / 32;
        b_rd[t][i] = A_STAGE + row * (BK / 2) + (((c >> 1) ^ ((row >> 3) & (KSP - 1))) << 4) +
                     (c & 1) * 8;
      }
    }
  }

#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int e = 0; e < ACC; ++e) acc[a][b][e] = 0;
  constexpr int PARAM_OFF = igemm_main_bytes<BM, BN, BK, STAGES>();
  float* P_B0 = reinterpret_cast<float*>(smem + PARAM_OFF);
  float* P_SC = P_B0 + BN;
  __half* P_BS = reinterpret_cast<__half*>(P_SC + BN);

  if constexpr (PHASED) {
    // fragment read offsets: the swizzle term depends on the lane's row within its 16-row tile
    // only, so every other tile of the wave is the same address plus an immediate
    const int a_lane = (wm * WTM + lrow) * BK, b_lane = A_STAGE + (wn * WTN + lrow) * BK;
    int fr[2];                             // 16-byte chunk of k-step ks for this lane, swizzled
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fr[ks] = ((ks * CPS + lkq) ^ swz<BK>(lrow)) << 4;
    v4i af[4][2], bf[2][2];
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
#if MIXDQ_ABLATE == 1
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) asm volatile("" ::"v"(bf[tn][ks]));
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) asm volatile("" ::"v"(af[tm][ks]));
      }
#else
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
            acc[nh * 2 + tn][mh * 4 + tm] =
                mfma(bf[tn][ks], af[tm][ks], acc[nh * 2 + tn][mh * 4 + tm]);
#endif
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_barrier" ::: "memory");
    };
    // tile 0's A0 and B0 have landed (its B1, A1 may fly) and are visible to every wave
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if (wid >= 4) asm volatile("s_barrier" ::: "memory");   // the second wave group: one barrier behind
    for (int kt = 0; kt < nk; ++kt) {
      const char* S0 = smem + (kt & 1) * STAGE;
      const int nb = (kt + 1) & 1, nkk = (kt + 1) * BK;
      // a unit is read one phase (two barriers) after the wait that retires it: both wave groups'
      // pieces have then been waited for in front of a barrier the reader has passed
      read_b(S0, 0); __builtin_amdgcn_sched_barrier(0); read_a(S0, 0);
      if (MIXDQ_ABLATE != 3) stage_unit(nb, nkk, 0);
#define MIXDQ_PH_WAIT() do { if (MIXDQ_ABLATE != 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } while (0)
      MIXDQ_PH_WAIT();                                      // B1 of this tile (phase 1)
      quadrant(0, 0);
      read_b(S0, 1);
      if (MIXDQ_ABLATE != 3) stage_unit(nb, nkk, 1);
      MIXDQ_PH_WAIT();                                      // A1 of this tile (phase 2)
      quadrant(0, 1);
      read_a(S0, 1);
      if (MIXDQ_ABLATE != 3) stage_unit(nb, nkk, 2);
      quadrant(1, 1);
      read_b(S0, 0);
      if (MIXDQ_ABLATE != 3) stage_unit(nb, nkk, 3);
      MIXDQ_PH_WAIT();                                      // A0, B0 of the next tile (its phase 0)
      quadrant(1, 0);
    }
    if (wid < 4) asm volatile("s_barrier" ::: "memory");    // the groups meet again
  }
  for (int kt0 = 0; !PHASED && kt0 < nk; kt0 += STAGES) {
#pragma unroll
    for (int s = 0; s < STAGES; ++s) {      // tile kt0 + s lives in buffer s (compile-time)
      const int kt = kt0 + s;
      if (kt < nk) {
        wait_tile();
        if (kt == 0) MIXDQ_STAMP_AT(2);
        const char* S0 = smem + s * STAGE;
        if constexpr (KS * (TM + TN) <= 16) {
          // small wave tiles are latency-bound: put every fragment read of the K-tile in flight
          // FIRST, issue the next stage's DMAs (address arithmetic) under the LDS latency, then
          // the MFMAs
          v4i af[KS][TM], bf[KS][TN];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#if MIXDQ_ABLATE == 2
#pragma unroll
            for (int t = 0; t < TM; ++t) { af[ks][t] = v4i{lane, ks, kt, t}; asm volatile("" : "+v"(af[ks][t])); }
#pragma unroll
            for (int t = 0; t < TN; ++t) { bf[ks][t] = v4i{lane, ks, kt, t}; asm volatile("" : "+v"(bf[ks][t])); }
#else
#pragma unroll
            for (int t = 0; t < TM; ++t)
              af[ks][t] = *reinterpret_cast<const v4i*>(S0 + a_rd[t][ks]);
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[ks][t] = load_w(S0, b_rd[t][ks]);
#endif
          }
#if MIXDQ_ABLATE != 3
          aq_slot((s + AD) % STAGES, [&](auto c) { a_load(c, (kt + AD) * BK); });
          stage((s + PRE) % STAGES, (kt + PRE) * BK);
#endif
#if MIXDQ_ABLATE == 1
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < TM; ++t) asm volatile("" ::"v"(af[ks][t]));
#pragma unroll
            for (int t = 0; t < TN; ++t) asm volatile("" ::"v"(bf[ks][t]));
          }
#else
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
              for (int b = 0; b < TM; ++b) acc[a][b] = mfma(bf[ks][a], af[ks][b], acc[a][b]);
#endif
          if constexpr (AQ) {                  // the NEXT K-tile's image, behind this one's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            aq_slot((s + 1) % STAGES, [&](auto c) { a_wait(c, std::false_type{}); a_convert(c, (s + 1) % STAGES); });
          }
          continue;
        }
#if MIXDQ_ABLATE != 3
        aq_slot((s + AD) % STAGES, [&](auto c) { a_load(c, (kt + AD) * BK); });
        stage((s + PRE) % STAGES, (kt + PRE) * BK);
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          v4i af[TM], bf[TN];
#if MIXDQ_ABLATE == 2
#pragma unroll
          for (int t = 0; t < TM; ++t) { af[t] = v4i{lane, ks, kt, t}; asm volatile("" : "+v"(af[t])); }
#pragma unroll
          for (int t = 0; t < TN; ++t) { bf[t] = v4i{lane, ks, kt, t}; asm volatile("" : "+v"(bf[t])); }
#else
#pragma unroll
          for (int t = 0; t < TM; ++t) af[t] = *reinterpret_cast<const v4i*>(S0 + a_rd[t][ks]);
#pragma unroll
          for (int t = 0; t < TN; ++t) bf[t] = load_w(S0, b_rd[t][ks]);
#endif
#if MIXDQ_ABLATE == 1
#pragma unroll
          for (int t = 0; t < TM; ++t) asm volatile("" ::"v"(af[t]));
#pragma unroll
          for (int t = 0; t < TN; ++t) asm volatile("" ::"v"(bf[t]));
#else
#pragma unroll
          for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) acc[a][b] = mfma(bf[a], af[b], acc[a][b]);
#endif
        }
        if constexpr (AQ) {                    // the NEXT K-tile's image, behind this one's MFMAs
          __builtin_amdgcn_sched_barrier(0);
          aq_slot((s + 1) % STAGES, [&](auto c) { a_wait(c, std::false_type{}); a_convert(c, (s + 1) % STAGES); });
        }
      }
    }
  }
  // the zero-page DMAs staged for tiles >= nk are still in flight: drain before LDS is reused
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::"v"(karg_warm));     // (the warm-up load's only use: nothing waits for it before this)
  MIXDQ_STAMP_AT(3);
  // ---- residual operand of the LARGE tiles (more than two 16-byte chunks per thread: not parked in
  //      registers across the main loop): every chunk of it is requested at once -- in front of the
  //      accumulator pass where the registers are there for it (the address unit is idle during that pass),
  //      behind it otherwise (the accumulators are dead by then).  (Read inside the store loop, one dependent memory round trip per iteration -- the
  //      compiler cannot move a load of `res` across a store to `D` -- the store phase of an (8192, 1280,
  //      1280) launch with a residual took 8.1-8.9 us of its 28 us: tools/stamp_report.py 8192 1280 1280
  //      --cfg 25 --res.)
  constexpr int ST_ITERS_ = (BM * (BN / 8) + 64 * WM * WN * KSPLIT - 1) / (64 * WM * WN * KSPLIT);
  constexpr bool RES_LATE = !RES_PRE && !ATT && !GROUPED && (ST_ITERS_ <= 10 || (PHASED && ST_ITERS_ <= 16));
  // EARLY: the requests are interleaved with the accumulator pass (one chunk every few quads: asked for in
  // one burst they are 80 KB through an address unit that takes ~1 KiB per 40 cycles -- every wave of the CU
  // stood in that queue for ~3 us before its pass could start; tools/stamp_report.py).  Needs the registers
  // for accumulators + chunks + the pass, and every wave in the pass (no k-split groups).
  constexpr bool RES_LATE_EARLY = RES_LATE && !PHASED && KSPLIT == 1 &&
                                  TN * TM * ACC + 4 * ST_ITERS_ + 60 <= REG_CAP;
  v4i res_late[RES_LATE ? ST_ITERS_ : 1];
  const bool res_late_on = RES_LATE && p.res != nullptr && (p.N & 7) == 0;
  const bool res_late_full = p.res_div == 1;
  auto request_residual_one = [&](int it) {   // `it`: a compile-time constant at every call site
    const int idx = min(tid + it * (64 * WM * WN * KSPLIT), BM * (BN / 8) - 1);
    const int row = idx / (BN / 8), cc = idx - row * (BN / 8);
    const int64_t m = min(m0 + row, p.M - 1);
    const int n = n0 + cc * 8 < p.N ? n0 + cc * 8 : 0;
    res_late[it] = *reinterpret_cast<const v4i*>(p.res + (res_late_full ? m : m / p.res_div) * p.N + n);
  };
  auto request_residual = [&]() {
    if constexpr (RES_LATE) {
      if (res_late_on) {
#pragma unroll
        for (int it = 0; it < ST_ITERS_; ++it) request_residual_one(it);
      }
    }
  };
  if (pre_on) {      // the epilogue vectors -> LDS (their own region: no one reads it before the barrier)
    v4f b0 = {0.f, 0.f, 0.f, 0.f}, sc = {0.f, 0.f, 0.f, 0.f};
    uint2 bs = make_uint2(0u, 0u);
    if (pre_in) {
      if constexpr (!F16) { b0 = pre_b0; sc = pre_sc; }
      if (has_bias) bs = pre_bs;
    }
    *reinterpret_cast<v4f*>(P_B0 + tid * 4) = b0;
    *reinterpret_cast<v4f*>(P_SC + tid * 4) = sc;
    *reinterpret_cast<uint2*>(P_BS + tid * 4) = bs;
  }

  // ---- epilogue: registers -> f16 tile in LDS -> whole-row 16-byte stores --------------------
  // (With residual chunks in flight the barriers of this pass are raw: __syncthreads() waits vmcnt(0) as well
  //  -- it would wait for the very loads that are meant to land under the pass.  What the barriers order here
  //  is LDS traffic only: the DMAs were drained above.)
  auto epi_barrier = [&]() {
    if constexpr (RES_LATE_EARLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  epi_barrier();     // every wave is done reading the stage buffers
  MIXDQ_STAMP_AT(4);
  if constexpr (KSPLIT > 1) {
    // groups 1.. park their partial accumulators (behind the fp16 tile's area), group 0 adds them
    constexpr int WREGS = TN * TM * ACC;          // accumulator registers of one wave
    static_assert(BM * CS_STRIDE + (KSPLIT - 1) * WM * WN * WREGS * 64 * 4 <=
                      igemm_main_bytes<BM, BN, BK, STAGES>(), "partials fit the stage buffers");
    using part_t = typename std::conditional<F16, float, int>::type;   // F16: fixed-order fp32 adds
    part_t* part = reinterpret_cast<part_t*>(smem + BM * CS_STRIDE) + ((wid % (WM * WN)) * WREGS * 64 + lane);
    constexpr int GROUP_INTS = WM * WN * WREGS * 64;
    if (kg != 0) {
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int e = 0; e < ACC; ++e)
            part[(kg - 1) * GROUP_INTS + ((a * TM + b) * ACC + e) * 64] = acc[a][b][e];
    }
    epi_barrier();
    if (kg == 0) {
#pragma unroll
      for (int g = 0; g < KSPLIT - 1; ++g)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int e = 0; e < ACC; ++e)
              acc[a][b][e] += part[g * GROUP_INTS + ((a * TM + b) * ACC + e) * 64];
    }
  }
  char* Cs = smem;
  const bool unfused = p.unfused != 0;
  const float zpv = zpv_early;                     // (padded convs: read at kernel entry, see there)
  // The accumulator -> fp16 pass is straight-line code per wave: the uniform choices (bias or not,
  // fused multiply-add or not) are taken ONCE, outside (MODE), the per-channel vectors come from LDS
  // (zero-filled past N, so there is no column test: a quad past N computes zeros into a part of the
  // tile that is never stored), and only a padded conv's BORDER pixels read anything from memory.
  // (Round 2 selected between the LDS row and the border-class row by POINTER: hipcc turned that
  // into one flat_load per register quad followed by vmcnt(0) -- twenty serial memory round trips,
  // 4.4 of the 26 us of the (1024, 10240, 1280) launch and ~1 us of every small GEMM; tools/
  // stamp_report.py.)
  // ---- GEMM + GEGLU + quantize, in registers.  The weight rows arrive as value|gate groups of 16
  //      ([v 0..15 | g 0..15 | v 16..31 | ...], include/mixdq_hip.h), so every 32-column MFMA tile (or
  //      pair of 16-column tiles) holds whole groups, and in the accumulator layout the lane that holds
  //      value columns c..c+3 of a row holds gate columns c+16..c+19 of it too: the fp16 tile never
  //      goes through LDS.  What LDS holds instead: the INT8 output tile (whole rows for the stores)
  //      and -- on the tiles that have their CU to themselves -- the GELU table (76 KB, DMA'd under the
  //      accumulator -> fp16 pass).  Every rounding point of the unfused chain (GEMM -> fp16, gelu ->
  //      fp16, product -> fp16, quantize) is kept: the INT8 tensor is the one mixdq_geglu_quantize
  //      produces from this GEMM's fp16 output.  ((8192, 10240, 1280) on the 256x256 tile: 117 us;
  //      142 with the fp16 tile staged through LDS and GELU computed, as round 2 did.)
  if constexpr (BN % 32 == 0 && WTN % 32 == 0 && !CONV && !F16 && !ATT && !GROUPED && !AQ && !LNQ) {
    if (p.Dq != nullptr) {
      constexpr int QS = BN / 2 + 16;             // INT8 tile row stride (bytes)
      constexpr bool TAB = igemm_gelu_table_fits<BM, BN, BK, STAGES>();
      constexpr int OQ = MT == 32 ? 2 * TN : TN / 2;   // output quads (4 consecutive channels) per row
      static_assert(WTN % 32 == 0 && (MT == 32 || TN % 2 == 0), "whole value|gate groups per wave");
      static_assert(BM * QS <= igemm_main_bytes<BM, BN, BK, STAGES>(), "INT8 tile fits");
      const char* Tb = smem + geglu_tile_bytes(BM, BN);
      if constexpr (TAB) {
        constexpr int PIECES = kGeluTabBytes / 1024;
        const char* src = reinterpret_cast<const char*>(g_gelu_tab);
#pragma unroll
        for (int j = 0; j < (PIECES + NWAVES - 1) / NWAVES; ++j) {
          const int q = wid + NWAVES * j;
          if (q < PIECES) glds16(src + q * 1024 + lane * 16, smem + geglu_tile_bytes(BM, BN) + q * 1024);
        }
      }
      const float s_inv = *p.g_sinv, zpq = *p.g_zp;
      // output quad oq of MFMA row-tile tm: tile-local value column (the gate is 16 further) and the
      // accumulator quads that hold them
      auto vcol = [&](int oq) {
        return MT == 32 ? wn * WTN + (oq >> 1) * 32 + 8 * (oq & 1) + 4 * lkq : wn * WTN + oq * 32 + 4 * lkq;
      };
      uint2 hv[TM][OQ], hg[TM][OQ];               // the wave's patch as packed fp16 quads
      auto to_regs = [&](auto mode_c, int tm) {   // MFMA row-tile tm of the patch
        constexpr int MODE = decltype(mode_c)::value;   // as to_tile below
        // The per-channel vectors of a BATCH of quads are read from LDS first, all of them, and only then
        // the batch is computed (a scheduling fence between the two): left to itself hipcc issued the three
        // reads of a quad, waited lgkmcnt(0), computed, read the next -- forty exposed LDS round trips per
        // lane in the 128x320 tile's pass (2.5 of the launch's 23 us, the other wave of the SIMD in the
        // same state).
        // (Batches of 4 quad-halves where the tile has its CU -- and so the registers -- to itself; the
        // tiles that share a CU keep one quad at a time: batched, they spilled.)
        constexpr int HB = (REG_CAP >= 192 && (MT == 16 || TN * TM * ACC <= 96)) ? 4 : 1;   // quad-halves per batch
#pragma unroll
        for (int h0 = 0; h0 < 2 * OQ; h0 += HB) {
          v4f b0[HB], sc[HB];
          v4h bsh[HB];
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            if (h0 + j >= 2 * OQ) break;
            const int oq = (h0 + j) >> 1, half = (h0 + j) & 1;
            const int nl = vcol(oq) + 16 * half;
            b0[j] = *reinterpret_cast<const v4f*>(P_B0 + nl);
            sc[j] = *reinterpret_cast<const v4f*>(P_SC + nl);
            if constexpr (MODE != 0) bsh[j] = *reinterpret_cast<const v4h*>(P_BS + nl);
          }
          if constexpr (HB > 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < HB; ++j) {
            if (h0 + j >= 2 * OQ) break;
            const int oq = (h0 + j) >> 1, half = (h0 + j) & 1;      // value quad, gate quad
            const int tn = MT == 32 ? (oq >> 1) : 2 * oq + half;
            const int e0 = MT == 32 ? 4 * ((oq & 1) + 2 * half) : 0;
            v4f b0q = b0[j], scq = sc[j];
            if constexpr (W4) { b0q = b0q * 16.0f; scq = scq * 0.0625f; }   // exact: the MFMA ran on 16*q
            v4f bs = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MODE != 0) bs = __builtin_convertvector(bsh[j], v4f);
            uint32_t packed[2];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              v2f x = {(float)acc[tn][tm][e0 + 2 * e2], (float)acc[tn][tm][e0 + 2 * e2 + 1]};
              const v2f b0e = {b0q[2 * e2], b0q[2 * e2 + 1]};
              const v2f sce = {scq[2 * e2], scq[2 * e2 + 1]};
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
          if (HB > 1 || (h0 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      };
      const bool mine = KSPLIT == 1 || kg == 0;   // k-split: group 0 holds the sums
      const int mode = !has_bias ? 0 : !unfused ? 1 : 2;
      auto to_regs_tm = [&](int tm) {
        if (mode == 0) to_regs(std::integral_constant<int, 0>{}, tm);
        else if (mode == 1) to_regs(std::integral_constant<int, 1>{}, tm);
        else to_regs(std::integral_constant<int, 2>{}, tm);
      };
      if constexpr (TAB) {                        // the whole patch first: the table is still in flight
        if (mine) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) to_regs_tm(tm);
        }
        MIXDQ_STAMP_AT(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                          // the table has landed, for every wave
      }
      MIXDQ_STAMP_AT(6);
      if (mine) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (!TAB) to_regs_tm(tm);     // no table to wait for: row-tile by row-tile
          const int ml = wm * WTM + tm * MT + lrow;
          uint32_t pk[OQ];
          auto quads = [&](auto how_c, auto unf_c) {
            constexpr int HOW = decltype(how_c)::value;
            constexpr bool UNF = decltype(unf_c)::value;
#pragma unroll
            for (int oq = 0; oq < OQ; ++oq) pk[oq] = geglu_quad<HOW, UNF>(hv[tm][oq], hg[tm][oq], Tb, s_inv, zpq);
          };
          if constexpr (TAB) {
            bool far = false;
#pragma unroll
            for (int oq = 0; oq < OQ; oq += 2)
              far |= geglu_any_far(hg[tm][oq].x, hg[tm][oq].y, hg[tm][oq + 1 < OQ ? oq + 1 : oq].x,
                                   hg[tm][oq + 1 < OQ ? oq + 1 : oq].y);
            if (__builtin_amdgcn_ballot_w64(far) == 0) {       // wave-uniform: the usual case
              if (unfused) quads(std::integral_constant<int, 0>{}, std::true_type{});
              else quads(std::integral_constant<int, 0>{}, std::false_type{});
            } else {
              if (unfused) quads(std::integral_constant<int, 1>{}, std::true_type{});
              else quads(std::integral_constant<int, 1>{}, std::false_type{});
            }
          } else {
            if (unfused) quads(std::integral_constant<int, 2>{}, std::true_type{});
            else quads(std::integral_constant<int, 2>{}, std::false_type{});
          }
#pragma unroll
          for (int oq = 0; oq < OQ; ++oq) {
            const int v = vcol(oq);                           // group start / 2 + offset in the group
            *reinterpret_cast<uint32_t*>(smem + ml * QS + ((v & ~31) >> 1) + (v & 15)) = pk[oq];
          }
        }
      }
      MIXDQ_STAMP_AT(10);
      __syncthreads();
      MIXDQ_STAMP_AT(11);
      const int Dh = p.N >> 1;
      const bool al16 = ((uintptr_t)p.Dq & 15) == 0;
      constexpr int CH = BN / 32;                 // 16-output chunks per tile row = its groups
      for (int idx = tid; idx < BM * CH; idx += NTHREADS) {
        const int row = idx / CH, cc = idx - row * CH;
        const int64_t m = m0 + row;
        if (m >= p.M || n0 + 32 * cc >= p.N) continue;        // N % 32 == 0: groups are whole
        const uint4 v = *reinterpret_cast<const uint4*>(smem + row * QS + cc * 16);
        int8_t* dst = p.Dq + m * Dh + (n0 >> 1) + cc * 16;
        if (al16) {
          *reinterpret_cast<uint4*>(dst) = v;
        } else {
          *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
          *reinterpret_cast<uint2*>(dst + 8) = make_uint2(v.z, v.w);
        }
      }
      MIXDQ_STAMP_AT(7);
      return;
    }
  }
  auto to_tile = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;   // 0: no bias, 1: bias (FMA / FP16 add), 2: bias, mul then add
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int ml = wm * WTM + tm * MT + lrow;
      // table mode (padded convs only): border class of this output pixel = its valid tap rectangle
      // [rlo,rhi]x[slo,shi]; interior pixels (the full window) use the row staged in LDS, border
      // pixels read theirs from the table
      const __attribute__((address_space(1))) float* b0row = nullptr;
      if constexpr (CONV) {
        if (use_table) {
          int64_t m = m0 + ml;
          if (m >= p.M) m = 0;
          const int pq = p.P * p.Q;
          const int rem = (int)(m % pq);
          const int pp = rem / p.Q, qq = rem - pp * p.Q;
          const int hb = pp * p.stride - p.pad, wb = qq * p.stride - p.pad;
          const int rlo = max(0, -hb), rhi = max(min(p.R - 1, p.H - 1 - hb), 0);
          const int slo = max(0, -wb), shi = max(min(p.S - 1, p.W - 1 - wb), 0);
          const int cls = ((min(rlo, p.R - 1) * p.R + rhi) * p.S + min(slo, p.S - 1)) * p.S + shi;
          if (cls != full_cls)
            b0row = (const __attribute__((address_space(1))) float*)(p.table + (int64_t)cls * p.N);
        }
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int g = 0; g < ACC / 4; ++g) {     // register quads: 4 consecutive output channels each
          if constexpr (RES_LATE_EARLY) {       // one residual chunk requested every QSTEP quads of the pass
            constexpr int NQUADS = TM * TN * (ACC / 4), QSTEP = NQUADS / ST_ITERS_ > 0 ? NQUADS / ST_ITERS_ : 1;
            const int qi = (tm * TN + tn) * (ACC / 4) + g;
            if (res_late_on && qi % QSTEP == 0 && qi / QSTEP < ST_ITERS_) request_residual_one(qi / QSTEP);
          }
          const int nl = wn * WTN + tn * MT + (MT == 32 ? 8 * g + 4 * lkq : 4 * lkq);
          v4f b0 = *reinterpret_cast<const v4f*>(P_B0 + nl);
          if constexpr (CONV) {
            if (b0row != nullptr && n0 + nl < p.N) {         // border pixels only (a real branch)
              b0 = *reinterpret_cast<const __attribute__((address_space(1))) v4f*>(b0row + n0 + nl);
              asm volatile("" : "+v"(b0));
            }
            if (use_table) b0 = b0 * zpv;                    // f32(sum of taps) * zp, one rounding
          }
          v4f sc = *reinterpret_cast<const v4f*>(P_SC + nl);
          if constexpr (W4) { b0 = b0 * 16.0f; sc = sc * 0.0625f; }   // exact: the MFMA ran on 16*q
          v4f bs = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MODE != 0)
            bs = __builtin_convertvector(*reinterpret_cast<const v4h*>(P_BS + nl), v4f);   // exact
          uint32_t packed[2];
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            // two outputs at a time: (f32(acc) - bias0) * scale [+ bias]
            v2f x = {(float)acc[tn][tm][4 * g + 2 * e2], (float)acc[tn][tm][4 * g + 2 * e2 + 1]};
            const v2f b0e = {b0[2 * e2], b0[2 * e2 + 1]};
            const v2f sce = {sc[2 * e2], sc[2 * e2 + 1]};
            const v2f bse = {bs[2 * e2], bs[2 * e2 + 1]};
            v2f r;
            if constexpr (F16) {
              r = MODE != 0 ? x + bse : x;                   // fp32 accumulator + bias, one rounding
            } else {
              x = x - b0e;
              if constexpr (MODE == 0) r = x * sce;
              else if constexpr (MODE == 2) r = x * sce + bse;   // -ffp-contract=off: mul, then add
              else r = __builtin_elementwise_fma(x, sce, bse);
            }
            asm("" : "+v"(r));   // keep the FP32 rounding: no fold into a single-rounding fma_mix
            const v2h h = __builtin_convertvector(r, v2h);    // v_cvt_pk_f16_f32, RNE
            packed[e2] = *reinterpret_cast<const uint32_t*>(&h);
          }
          *reinterpret_cast<uint2*>(Cs + ml * CS_STRIDE + nl * 2) = make_uint2(packed[0], packed[1]);
          // four quads' worth of LDS reads in flight at a time: without the fence the scheduler
          // hoists every quad's reads to the top and the pass costs ~100 registers (spills, or the
          // co-resident workgroup)
          if ((tn * (ACC / 4) + g) % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if (KSPLIT == 1 || kg == 0) {                     // group 0 holds the sums
    if (!has_bias) to_tile(std::integral_constant<int, 0>{});
    else if (F16 || !unfused) to_tile(std::integral_constant<int, 1>{});
    else to_tile(std::integral_constant<int, 2>{});
  }
  MIXDQ_STAMP_AT(5);
  if constexpr (RES_LATE && !RES_LATE_EARLY) request_residual();
  if constexpr (RES_LATE_EARLY) {
    constexpr int NQUADS = TM * TN * (ACC / 4), QSTEP = NQUADS / ST_ITERS_ > 0 ? NQUADS / ST_ITERS_ : 1;
    constexpr int COVERED = (NQUADS + QSTEP - 1) / QSTEP < ST_ITERS_ ? (NQUADS + QSTEP - 1) / QSTEP : ST_ITERS_;
    if (res_late_on) {
#pragma unroll
      for (int it = COVERED; it < ST_ITERS_; ++it) request_residual_one(it);
    }
  }
  __syncthreads();
  MIXDQ_STAMP_AT(6);
  if constexpr (ATT) {
    // ---- cross-attention on the staged tile: wave w < 4 owns head (w >> 1) of the pair and 32 of
    //      the 64 query rows; arithmetic and order are those of attn_fwd_kernel (csrc/attention.hip)
    //      for two key tiles, so the result is bit-identical to to_q followed by that kernel.
    const int l32 = lane & 31, hh = lane >> 5;
    const int hl = wid >> 1, rg = wid & 1;
    v8h qf[4];
    if (wid < 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const v8h*>(Cs + (rg * 32 + l32) * CS_STRIDE +
                                               (hl * kHeadDim + ks * 16 + hh * 8) * 2);
    }
    __syncthreads();                   // the fp16 tile is consumed: its LDS becomes output staging
    if (wid >= 4) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned kv0 = lds0 + ATT_OFF + hl * (2 * kStageBytes);
    unsigned k_a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_a[ks] = kv0 + l32 * kRow + (((2 * ks + hh) ^ ((l32 >> 1) & 7)) << 4);
    const int q4 = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
    const int v_rd0 = kTileBytes + (4 * hh + q4) * kRow +
                      (((2 * g16 + (pp >> 1)) ^ (((q4 >> 1) & 1) << 2)) << 4) + 8 * (pp & 1);
    const unsigned v_a0 = kv0 + v_rd0, v_a1 = kv0 + (v_rd0 ^ 64);
    const int ntiles = (p.att_tkv + kKeys - 1) / kKeys;
    v16f o[2], lsum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; lsum[i] = 0.f; }
    float m_i = -INFINITY;
    const float c = p.att_scale_log2;
    v8h ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (_Float16)1.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= ntiles) break;
      VFrag vf[2][2][2];
      {
        const unsigned a0 = v_a0 + t * kStageBytes, a1 = v_a1 + t * kStageBytes;
        tr_read2_imm<0 * kRow>(vf[0][0][0], a0);
        tr_read2_imm<0 * kRow>(vf[0][0][1], a1);
        tr_read2_imm<16 * kRow>(vf[0][1][0], a0);
        tr_read2_imm<16 * kRow>(vf[0][1][1], a1);
        tr_read2_imm<32 * kRow>(vf[1][0][0], a0);
        tr_read2_imm<32 * kRow>(vf[1][0][1], a1);
        tr_read2_imm<48 * kRow>(vf[1][1][0], a0);
        tr_read2_imm<48 * kRow>(vf[1][1][1], a1);
      }
      v16f sc[2];
      {
        v8h kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            kf[kb][ks] = *(const __attribute__((address_space(3))) v8h*)(size_t)(
                k_a[ks] + (t * kStageBytes + kb * 32 * kRow));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            sc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], sc[kb], 0, 0, 0);
        }
      }
      if (t == ntiles - 1 && (p.att_tkv & (kKeys - 1)) != 0) {   // mask the absent keys
        const int lim = p.att_tkv - t * kKeys - 4 * hh;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * kb + 8 * (r >> 2) + (r & 3) >= lim) sc[kb][r] = -INFINITY;
      }
      float mx = sc[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
      mx = half_max(mx);
      const float m_new = fmaxf(m_i, mx);
      const bool grew = m_new > m_i;
      const float mc = m_new * c;
      v8h pf[2][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pf[kb][r >> 3][r & 7] =
              (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], c, -mc));
      if (__builtin_amdgcn_ballot_w64(grew)) {
        const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c);
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        lsum[0] *= alpha;
      }
      m_i = m_new;
      s_waitcnt_lgkm0();
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int db = 0; db < 2; ++db)
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][u][db].h, pf[kb][u], o[db], 0, 0, 0);
          lsum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[kb][u], lsum, 0, 0, 0);
        }
    }
    // normalise, stage through LDS (per wave), store whole 64-column head rows
    const float inv = 1.f / lsum[0];
    char* Os = smem + wid * (32 * kORow);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        v4h w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (_Float16)(o[db][4 * g + j] * inv);
        *reinterpret_cast<v4h*>(Os + l32 * kORow + (32 * db + 8 * g + 4 * hh) * 2) = w;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging: no block barrier
    const bool quant = p.att_sinv != nullptr;
    const float s_inv = att_sinv_early, zpq = att_zp_early;    // (read at kernel entry: see below)
    const bool unf = p.unfused != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = lane + 64 * i, row = id >> 3, ch = id & 7;
      const int64_t m = m0 + rg * 32 + row;
      if (m >= p.M) continue;
      const uint4 w = *reinterpret_cast<const uint4*>(Os + row * kORow + ch * 16);
      const int64_t off = m * p.N + n0 + hl * kHeadDim + ch * 8;
      if (!quant) {
        *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(p.att_out) + off) = w;
      } else {
        const __half* hv = reinterpret_cast<const __half*>(&w);
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __half2float(hv[j]);
        *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(p.att_out) + off) =
            unf ? quantize_pack8<true>(x, s_inv, zpq) : quantize_pack8<false>(x, s_inv, zpq);
      }
    }
    return;
  }
  constexpr int CPRO = BN / 8;   // 16-byte chunks per output row of the tile
  const bool n8 = (p.N & 7) == 0;
  const bool identity_rows = p.grp_rows <= 0;
  const bool res_full = p.res != nullptr && p.res_div == 1;
  // thread -> (row, chunk): chunk-fastest, so the lanes of a wave write whole output rows
  constexpr int ST_ITERS = (BM * CPRO + NTHREADS - 1) / NTHREADS;
  static_assert(ST_ITERS == RES_ITERS, "the residual was requested with this loop's mapping");
  auto store_chunk = [&](auto it_c, int idx) {
    constexpr int it = decltype(it_c)::value;     // compile-time slot of the requested residual
    const int row = idx / CPRO, cc = idx - row * CPRO;
    const int n = n0 + cc * 8;
    const int64_t m = m0 + row;
    if (m >= p.M || n >= p.N) return;
    int64_t drow = m;
    if (!identity_rows) {
      const int64_t gq = m / p.grp_rows;
      drow = gq * p.grp_stride + p.grp_off + (m - gq * p.grp_rows);
    }
    uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_STRIDE + cc * 16);
    if (p.res != nullptr) {
      const int64_t rrow = res_full ? m : m / p.res_div;
      const __half* rp = p.res + rrow * p.N + n;
      uint32_t rw[4];
      if (n8) {
        uint4 r;
        if constexpr (RES_PRE) r = res_pre[it];     // requested before the main loop
        else if constexpr (RES_LATE) r = __builtin_bit_cast(uint4, res_late[it]);   // requested around the accumulator pass
        else r = *reinterpret_cast<const uint4*>(rp);
        rw[0] = r.x; rw[1] = r.y; rw[2] = r.z; rw[3] = r.w;
      } else {
        const uint2 r0 = *reinterpret_cast<const uint2*>(rp);
        rw[0] = r0.x; rw[1] = r0.y; rw[2] = 0; rw[3] = 0;
        if (n + 8 <= p.N) {
          const uint2 r1 = *reinterpret_cast<const uint2*>(rp + 4);
          rw[2] = r1.x; rw[3] = r1.y;
        }
      }
      uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
      for (int e = 0; e < 4; ++e) vw[e] = add_f16x2(vw[e], rw[e]);
    }
    if constexpr (LNQ) {    // the FINAL value (residual added) back into the tile: the LayerNorm's input; the
      *reinterpret_cast<uint4*>(Cs + row * CS_STRIDE + cc * 16) = v;   // rows leave for D behind the records
      // ... and the 16-column group's statistics while the chunk is in registers: a chunk is half a group, its
      // other half sits in the neighbouring lane (chunks run along the lanes; a pair never straddles a wave)
      float f[8];
      {
        const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          __half_raw r;
          r.x = (unsigned short)((j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xffffu));
          f[j] = __half2float(__half(r));
        }
      }
      float s8 = f[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) s8 = __fadd_rn(s8, f[j]);
      const float s1 = __fadd_rn(s8, ln_dpp<0xB1>(s8));
      const float mg = __fmul_rn(s1, 0.0625f);
      float q8 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = __fsub_rn(f[j], mg);
        q8 = __builtin_fmaf(d, d, q8);
      }
      const float m2 = __fadd_rn(q8, ln_dpp<0xB1>(q8));
      if ((cc & 1) == 0)
        *reinterpret_cast<float2*>(smem + LN_GRP_OFF + (row * (BN / 16) + (cc >> 1)) * 8) = make_float2(s1, m2);
      return;
    }
    __half* dst = p.D + drow * p.N + n;
#if MIXDQ_ABLATE == 4
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u)      // never true in practice: stores elided
#endif
    if (n8) {
      {
        // non-temporal: the output is a stream this launch never re-reads, it should not push
        // the operand panels out of the XCD's L2 ((32768, 1920, 640): 73 vs 86 us, (8192, 10240,
        // 1280): 150 vs 160 us; neutral at batch 1)
        const v4i vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        // global_store_dwordx4 ... nt, through the builtin: the compiler then counts the store in vmcnt and pads
        // the gfx950 store-data hazard (a store of more than 64 bits reads its data registers up to two wait
        // states after issue).  As inline asm it did neither: a VALU write of the first data register right
        // behind the asm was picked up by the store (tests/test_ops_gpu.py halo cases, round 4).
        __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
      }
    } else {   // N % 8 == 4: rows are only 8-byte aligned
      *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
      if (n + 8 <= p.N) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(v.z, v.w);
    }
  };
  if constexpr (RES_PRE) {
    static_assert(ST_ITERS <= 2, "one compile-time slot per requested residual chunk");
    if (tid < BM * CPRO) store_chunk(std::integral_constant<int, 0>{}, tid);
    if constexpr (ST_ITERS == 2)
      if (tid + NTHREADS < BM * CPRO) store_chunk(std::integral_constant<int, 1>{}, tid + NTHREADS);
  } else if constexpr (RES_LATE) {
    static_assert(ST_ITERS == ST_ITERS_, "the residual was requested with this loop's mapping");
    if (res_late_on) {      // compile-time slots of the requested residual chunks: the loop is unrolled
      igemm_unrolled<ST_ITERS>([&](auto it_c) {
        const int idx = tid + decltype(it_c)::value * NTHREADS;
        if (idx < BM * CPRO) store_chunk(it_c, idx);
      });
    } else {
      for (int idx = tid; idx < BM * CPRO; idx += NTHREADS) store_chunk(std::integral_constant<int, 0>{}, idx);
    }
  } else {
    for (int idx = tid; idx < BM * CPRO; idx += NTHREADS) store_chunk(std::integral_constant<int, 0>{}, idx);
  }
  MIXDQ_STAMP_AT(7);
  if constexpr (LNQ) {
    // ---- LayerNorm + quantize of the rows this launch just produced (round 5).  A row's statistics need all
    //      N columns and a tile holds BN of them -- exactly one UNIT of the LayerNorm specification
    //      (oracle/mixdq_oracle.c: BN / 16 groups folded left to right; N / BN <= 16 units combined by a balanced
    //      tree), so every tile publishes ONE 16-byte record per row: {sum, centred sum of squares, launch tag},
    //      and every tile reads the N / BN records of each of its rows, combines them itself and normalises +
    //      quantizes the BN columns it still holds in LDS.  No counter and no barrier between the tiles: a record
    //      is valid when its tag is this launch's (the workspace's epoch + 1; the last workgroup to leave bumps
    //      the epoch), the reader simply polls the record.  Visibility across CUs / XCDs: records are written
    //      and read with 16-byte sc1 (write-through / L1-bypassing) accesses, one granule each (observed
    //      untorn on gfx950, MI355X_MICROARCH.md "R2's granule needs no ordering at all").  All tiles of a row
    //      block are resident at once (the launcher: one workgroup per CU, one round).  A first form -- per-group
    //      8-byte partials, an arrival counter per row block -- cost three trips through the memory side and 5 K
    //      8-byte loads per workgroup: the step got 1.1 ms SLOWER than with a LayerNorm launch of its own
    //      (profiles/r05_ln_in_gemm.txt).  Replaces one ln_quant_kernel launch per LayerNorm of the fused graph.
    constexpr int GPT = BN / 16;                     // 16-column groups per tile row = groups per unit
    static_assert(GPT <= 8, "one 8-lane group computes a row's unit record");
    const int U = p.tiles_n;                         // units per row (the launcher: N == U * BN, U <= 16)
    const float n_u = (float)BN;
    const float inv_nu = 1.0f / n_u, inv_c = 1.0f / (float)p.N;    // (the specification multiplies by these)
    const int tag = ln_epoch + 1;
    float* ln_mr = reinterpret_cast<float*>(smem + BM * CS_STRIDE);     // [BM][2]: mean, rstd (behind the tile)
    static_assert(BM * CS_STRIDE + BM * 8 <= igemm_main_bytes<BM, BN, BK, STAGES>(), "row table fits");
    // Two copies of every record: A, written with a PLAIN store -- it stays in the writer's XCD L2, where the
    // other tiles of the row block find it with L1-bypassing (sc1) loads after one L2 round trip IF they run on
    // the same XCD (the tile map above makes that the usual case: workgroups b and b + 8 share an XCD -- an
    // observation, not a promise of the runtime); and B, written through to the memory side (sc1), which every
    // reader on every XCD sees.  A reader polls A a few times, then B: never wrong, fast when placed as expected.
    uint4* recs = reinterpret_cast<uint4*>(p.ln_part);
    uint4* recs_b = recs + p.M * 16;
    __syncthreads();                                 // the final fp16 tile is complete in Cs
    MIXDQ_STAMP_AT(6);
    auto halves = [](const uint4& w, float (&f)[8]) {
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __half_raw r;
        r.x = (unsigned short)((j & 1) ? (ww[j >> 1] >> 16) : (ww[j >> 1] & 0xffffu));
        f[j] = __half2float(__half(r));
      }
    };
    // the row's unit record from its GPT group statistics (computed from registers in the pass above): one
    // thread per row, groups left to right, Chan's combination about the unit mean
    for (int row = tid; row < BM; row += NTHREADS) {
      const float2* gs = reinterpret_cast<const float2*>(smem + LN_GRP_OFF) + row * GPT;
      float2 g[GPT];
#pragma unroll
      for (int k = 0; k < GPT; ++k) g[k] = gs[k];
      float s1u = g[0].x;
#pragma unroll
      for (int k = 1; k < GPT; ++k) s1u = __fadd_rn(s1u, g[k].x);
      const float mu = __fmul_rn(s1u, inv_nu);
      float m2u = 0.f;
#pragma unroll
      for (int k = 0; k < GPT; ++k) {
        const float e = __fsub_rn(__fmul_rn(g[k].x, 0.0625f), mu);
        const float c = __builtin_fmaf(__fmul_rn(e, 16.0f), e, g[k].y);
        m2u = k == 0 ? c : __fadd_rn(m2u, c);
      }
      if (m0 + row < p.M) {
        const v4i rec = {(int)__float_as_uint(s1u), (int)__float_as_uint(m2u), tag, 0};
        const uint4* dst = recs + ((m0 + row) * U + tile_n);
        const uint4* dst_b = recs_b + ((m0 + row) * U + tile_n);
        asm volatile("global_store_dwordx4 %0, %2, off\n\tglobal_store_dwordx4 %1, %2, off sc1\n\ts_nop 1"
                     ::"v"(dst), "v"(dst_b), "v"(rec) : "memory");
      }
    }
    // the output rows leave now, under the records' way to the other tiles (in front of them they put ~1 us of
    // store traffic between the last MFMA and the first record)
    for (int idx2 = tid; idx2 < BM * CPRO; idx2 += NTHREADS) {
      const int row = idx2 / CPRO, cc = idx2 - row * CPRO;
      if (m0 + row >= p.M) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CS_STRIDE + cc * 16);
      const v4i vv = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
      __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(p.D + (m0 + row) * p.N + n0 + cc * 8));
    }
    MIXDQ_STAMP_AT(10);
    // gamma / beta of this thread's chunks and the quantizers' scalars: requested now, they land during the exchange
    uint4 gmv[ST_ITERS], btv[ST_ITERS];
#pragma unroll
    for (int it = 0; it < ST_ITERS; ++it) {
      const int idx2 = min(tid + it * NTHREADS, BM * CPRO - 1);
      const int cc = idx2 % CPRO;
      gmv[it] = *reinterpret_cast<const uint4*>(p.ln_gamma + n0 + cc * 8);
      btv[it] = *reinterpret_cast<const uint4*>(p.ln_beta + n0 + cc * 8);
    }
    float qs[3] = {0.f, 0.f, 0.f}, qz[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < p.ln_nq) { qs[k] = *p.ln_sinv[k]; qz[k] = *p.ln_zp[k]; }
    // ---- the rows' U records: 16 lanes per row, lane u polls unit u's record until it carries this launch's
    //      tag.  A thread's RPT rows are requested together (one trip through the memory side, not RPT).
    {
      constexpr int RPT = BM / (NTHREADS / 16);      // rows per thread: 2 (64-row tiles), 4 (128-row tiles)
      static_assert(RPT == 2 || RPT == 4, "poll loads are written for 2 or 4 rows per thread");
      const int u = tid & 15;
      bool live[RPT], ok[RPT];
      const uint4* src[RPT];
      v4i rec[RPT];
      const int kTryLocal = p.ln_local ? 12 : 0;     // polls of the L2-resident copy before the written-through one
      int64_t rec_i[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int row = (tid >> 4) + r * (NTHREADS / 16);
        live[r] = u < U && m0 + row < p.M;
        rec_i[r] = (m0 + (m0 + row < p.M ? row : 0)) * U + (u < U ? u : 0);
        src[r] = recs + rec_i[r];
        ok[r] = !live[r];
      }
      for (int spins = 0; spins < (1 << 18); ++spins) {
        if (spins == kTryLocal) {
#pragma unroll
          for (int r = 0; r < RPT; ++r) src[r] = recs_b + rec_i[r];
        }
        if constexpr (RPT == 2)
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(rec[0]), "=&v"(rec[1]) : "v"(src[0]), "v"(src[1]) : "memory");
        else
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                       "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(rec[0]), "=&v"(rec[1]), "=&v"(rec[2]), "=&v"(rec[3])
                       : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]) : "memory");
        bool all_ok = true;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          ok[r] = !live[r] || rec[r][2] == tag;
          all_ok = all_ok && ok[r];
        }
        if (__builtin_amdgcn_ballot_w64(!all_ok) == 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      MIXDQ_STAMP_AT(11);
      {   // a record that never arrived: besides poisoning the row (below), leave a STICKY error word in the workspace
        bool lost = false;      // (ln_cnt[2]: never cleared by a launch; mixdq_qlinear_ln_status reads it)
#pragma unroll
        for (int r = 0; r < RPT; ++r) lost = lost || (live[r] && !ok[r]);
        if (__builtin_amdgcn_ballot_w64(lost) != 0 && lane == 0)
          __hip_atomic_store(p.ln_cnt + 2, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int row = (tid >> 4) + r * (NTHREADS / 16);
        // (a record that never arrived -- a lost workgroup; bounded, not a hang -- poisons the row: NaN out)
        const float s1u = live[r] ? (ok[r] ? __uint_as_float((unsigned)rec[r][0]) : __uint_as_float(0x7fc00000u)) : 0.f;
        const float m2u = live[r] ? __uint_as_float((unsigned)rec[r][1]) : 0.f;
        const float mean = __fmul_rn(ln_row_tree(s1u, U), inv_c);
        const float e = __fsub_rn(__fmul_rn(s1u, inv_nu), mean);
        const float t2 = ln_row_tree(__builtin_fmaf(__fmul_rn(e, n_u), e, m2u), U);
        const float rstd = 1.0f / sqrtf(__fadd_rn(__fmul_rn(t2, inv_c), p.ln_eps));
        if (u == 0) { ln_mr[2 * row] = mean; ln_mr[2 * row + 1] = rstd; }
      }
    }
    __syncthreads();
    MIXDQ_STAMP_AT(14);
    // ---- normalise + quantize the tile's own columns, whole 8-column chunks per thread
    const bool unf = p.unfused != 0;
    igemm_unrolled<ST_ITERS>([&](auto it_c) {
      constexpr int it = decltype(it_c)::value;
      const int idx2 = tid + it * NTHREADS;
      if (idx2 >= BM * CPRO) return;
      const int row = idx2 / CPRO, cc = idx2 - row * CPRO;
      const int64_t m = m0 + row;
      if (m >= p.M) return;
      float x[8], gm[8], bt[8];
      halves(*reinterpret_cast<const uint4*>(Cs + row * CS_STRIDE + cc * 16), x);
      halves(gmv[it], gm);
      halves(btv[it], bt);
      const float mean = ln_mr[2 * row], rstd = ln_mr[2 * row + 1];
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float nrm = __fmul_rn(__fsub_rn(x[j], mean), rstd);
        y[j] = __half2float(f32_to_f16_rn(__builtin_fmaf(nrm, gm[j], bt[j])));
      }
      const int64_t off = m * p.N + n0 + cc * 8;
      if (p.ln_h != nullptr) {
        uint32_t hw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hw[j] = (uint32_t)__half_as_ushort(f32_to_f16_rn(y[2 * j])) |
                  ((uint32_t)__half_as_ushort(f32_to_f16_rn(y[2 * j + 1])) << 16);
        *reinterpret_cast<uint4*>(p.ln_h + off) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k >= p.ln_nq) break;
        *reinterpret_cast<uint2*>(p.ln_q[k] + off) =
            unf ? quantize_pack8<true>(y, qs[k], qz[k]) : quantize_pack8<false>(y, qs[k], qz[k]);
      }
    });
    MIXDQ_STAMP_AT(15);
    // ---- departures: the last workgroup of the launch to get here bumps the epoch (the next launch's tag)
    if (tid == 0) {
      if (__hip_atomic_fetch_add(p.ln_cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nwg - 1) {
        __hip_atomic_store(p.ln_cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.ln_cnt, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// m-tiles per super-row of the blockIdx -> tile map: 8 (super-rows of 1, 2, 4, 16 measured equal or worse
// on the UNet's shapes; MIXDQ_IGEMM_GM=<1..64> overrides it for A/B runs)
inline int tile_map_gm() {
  static const int gm = [] {
    const char* e = getenv("MIXDQ_IGEMM_GM");
    const int v = e ? atoi(e) : 8;
    return v >= 1 && v <= 64 ? v : 8;
  }();
  return gm;
}

template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool CONV, bool FAST, bool W4, int KSPLIT,
          int MT, bool F16 = false, bool PHASED = false, bool GROUPED = false, bool AQ = false, bool LNQ = false>
int launch_kernel(IgemmParams& p, hipStream_t stream) {
  // (LNQ: the LDS request is padded past half a CU's LDS, so that no two workgroups share a CU: the tiles of a
  //  row block wait for each other, and the cross-CU hand-off form they use is the one-workgroup-per-CU one)
  constexpr int SMEM = LNQ && igemm_smem_bytes<BM, BN, BK, STAGES>() < 82 * 1024
                           ? 82 * 1024 : igemm_smem_bytes<BM, BN, BK, STAGES>();
  static_assert(SMEM <= 160 * 1024, "LDS is 160 KiB per CU");
  if (p.Dq != nullptr && (BN % 32 != 0 || (BN / WN) % 32 != 0))        // whole value|gate groups per tile, per wave
    return MIXDQ_ERR_GEGLU_SHAPE;
  if constexpr (SMEM > 64 * 1024) {   // opt in to > 64 KiB of dynamic LDS, once per instantiation and device
    static bool seen[64] = {};
    if (const int st = lds_opt_in(
            reinterpret_cast<const void*>(
                &igemm_kernel<BM, BN, BK, STAGES, WM, WN, CONV, FAST, W4, KSPLIT, MT, F16, false, PHASED, GROUPED, AQ, LNQ>),
            SMEM, seen))
      return st;
  }
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (p.N + BN - 1) / BN;
  p.gm = tile_map_gm();
  const int64_t grid = (int64_t)p.tiles_m * p.tiles_n;
  if (grid <= 0 || grid > 0x7fffffff || p.tiles_m >= (1 << 24)) return MIXDQ_ERR_INVALID_ARG;
  const int ny = GROUPED ? p.ngroups_launch : 1;
  if (GROUPED != (p.groups != nullptr)) return MIXDQ_ERR_INVALID_ARG;
  if constexpr (LNQ) {   // every tile of a row block must be resident at once: one workgroup per CU, one round
    // ... and a column tile must be exactly one unit of the LayerNorm's reduction order
    int units = 1;
    while (units < 16 && (p.N / 16) % (2 * units) == 0) units *= 2;
    if (grid > kNumCU || p.N % BN != 0 || p.N / BN != units) return MIXDQ_ERR_SHAPE;
    // ... on THIS device: its real CU count (a partitioned CPX / DPX mode or a smaller part has fewer than 256)
    // and at least one workgroup of this kernel per CU by the runtime's own occupancy answer -- a grid that is not
    // wholly resident would wait for records of workgroups that cannot start until a waiter leaves.
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MIXDQ_ERR_LAUNCH;
    if (cus[dev] == 0) {
      int n = 0, per_cu = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return MIXDQ_ERR_LAUNCH;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
              &per_cu, igemm_kernel<BM, BN, BK, STAGES, WM, WN, CONV, FAST, W4, KSPLIT, MT, F16, false, PHASED, GROUPED, AQ, LNQ>,
              64 * WM * WN * KSPLIT, SMEM) != hipSuccess)
        return MIXDQ_ERR_LAUNCH;
      cus[dev] = per_cu >= 1 ? n : -1;
    }
    if (cus[dev] < 0 || grid > cus[dev]) return MIXDQ_ERR_SHAPE;
  }
  igemm_kernel<BM, BN, BK, STAGES, WM, WN, CONV, FAST, W4, KSPLIT, MT, F16, false, PHASED, GROUPED, AQ, LNQ>
      <<<dim3((unsigned)grid, (unsigned)ny), 64 * WM * WN * KSPLIT, SMEM, stream>>>(MIXDQ_IGEMM_HEAD_ARGS(p) p);
  return launch_status();
}

template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool CONV, bool W4, int KSPLIT, int MT,
          bool F16 = false, bool PHASED = false, bool GROUPED = false>
int launch_tile(IgemmParams& p, hipStream_t stream) {
  if constexpr (!CONV) {
    const bool fits32 = (uint64_t)p.M * (uint64_t)p.Ktot < (1ull << 32) &&
                        (uint64_t)p.N * (uint64_t)p.Ktot < (1ull << 32);
    if (p.Ktot % BK == 0 && fits32) {
      if constexpr (PHASED && !W4 && !F16)
        return launch_kernel<BM, BN, BK, STAGES, WM, WN, false, true, false, KSPLIT, MT, false, true>(p, stream);
      else if constexpr (!PHASED)
        return launch_kernel<BM, BN, BK, STAGES, WM, WN, false, true, W4, KSPLIT, MT, F16, false, GROUPED>(p, stream);
    }
  }
  if constexpr (PHASED)   // convs, packed weights, K tails: the same tile on the one-phase loop
    return launch_kernel<BM, BN, BK, STAGES, 4, 2, CONV, false, W4, 1, 32, F16>(p, stream);
  else
    return launch_kernel<BM, BN, BK, STAGES, WM, WN, CONV, false, W4, KSPLIT, MT, F16, false, GROUPED>(p, stream);
}

// AQ launches: the Linear fast path only -- whole K-tiles and 32-bit byte offsets into the FP16 operand
// (its last addressed row starts below 4 GiB).  Anything else: MIXDQ_ERR_SHAPE; the caller then runs the
// reference's two launches (quantize, GEMM).
template <int BM, int BN, int BK, int STAGES, int WM, int WN, bool W4, int KSPLIT, int MT>
int launch_tile_aq(IgemmParams& p, hipStream_t stream) {
  const uint64_t rows = p.a_rowmap && p.grp_rows > 0
                            ? ((uint64_t)(p.M + p.grp_rows - 1) / p.grp_rows) * (uint64_t)p.grp_stride + p.grp_off
                            : (uint64_t)p.M;
  const bool fits32 = rows * (uint64_t)p.a_ld * 2 + 2 * (uint64_t)p.Ktot < (1ull << 32) &&
                      (uint64_t)p.N * (uint64_t)p.Ktot < (1ull << 32);
  if (p.Ktot % BK != 0 || !fits32) return MIXDQ_ERR_SHAPE;
  return launch_kernel<BM, BN, BK, STAGES, WM, WN, false, true, W4, KSPLIT, MT, false, false, false, true>(p, stream);
}

// LNQ launches (GEMM + residual + LayerNorm + quantize): the Linear fast path on an exact-fit tile.
template <int BM, int BN, int BK, int STAGES, int WM, int WN, int KSPLIT, int MT>
int launch_tile_ln(IgemmParams& p, hipStream_t stream) {
  const bool fits32 = (uint64_t)p.M * (uint64_t)p.Ktot < (1ull << 32) &&
                      (uint64_t)p.N * (uint64_t)p.Ktot < (1ull << 32);
  if (p.Ktot % BK != 0 || !fits32) return MIXDQ_ERR_SHAPE;
  return launch_kernel<BM, BN, BK, STAGES, WM, WN, false, true, false, KSPLIT, MT, false, false, false, false, true>(p, stream);
}

}  // namespace
}  // namespace mixdq
