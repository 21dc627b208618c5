// a1 + a2 in one launch: INT8 GEMM whose activation operand arrives as FP16 and is quantized in the
// staging path of igemm_kernel (AQ instantiations; csrc/igemm_kernel.h).  Replaces the reference's
// quantize launch in front of every QuantizedLinear / 1x1 QuantizedConv2d (nn/Linear.py:162-176,
// nn/Conv2d.py:294-311: quant_op -> qlinear / qconv2d) behind the module-swap API.
//
// Its own translation unit: the AQ family is compiled beside the INT8-operand family of igemm.hip.
#include "igemm_kernel.h"

namespace mixdq {
namespace {

// Tile configurations the AQ family is built for: the ones the automatic rule picks for the UNet's Linear
// shapes (ids and parameters are those of MIXDQ_IGEMM_CONFIGS in igemm.hip).
#define MIXDQ_AQ_CONFIGS(X)               \
  X(4, 64, 64, 128, 3, 2, 2, 1, 32)       \
  X(13, 256, 128, 64, 3, 4, 2, 1, 32)     \
  X(27, 128, 320, 128, 2, 8, 2, 1, 16)    \
  X(28, 128, 320, 128, 2, 4, 4, 1, 16)    \
  X(35, 128, 128, 64, 3, 4, 2, 1, 32)     \
  X(37, 64, 64, 128, 3, 2, 2, 2, 32)      \
  X(41, 64, 128, 128, 3, 2, 4, 1, 32)     \
  X(44, 128, 80, 128, 3, 4, 1, 2, 16)     \
  X(45, 64, 80, 128, 4, 4, 1, 2, 16)      \
  X(56, 64, 80, 128, 6, 4, 1, 2, 16)

struct AqCfg { int id, bm, bn, bk; };
constexpr AqCfg kAqCfgs[] = {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) {ID, BM, BN, BK},
    MIXDQ_AQ_CONFIGS(X)
#undef X
};

// The INT8 rule (igemm.hip: mixdq_igemm_select_id / _w4) mapped into the list above.
int select_aq(int64_t M, int N, int K, bool w4) {
  int c = w4 ? mixdq_igemm_select_id_w4(M, N, K, K) : mixdq_igemm_select_id(M, N, K, K);
  switch (c) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) case ID:
    MIXDQ_AQ_CONFIGS(X)
#undef X
      return c;
    case 20: case 18: case 14: case 70: return 13;   // 256-row tiles -> 256x128x64
    case 3: return 35;
    case 25: return 27;
    case 42: case 43: return 56;
    default: return c <= 0 ? -1 : 35;
  }
}

int aq_bk(int cfg) {
  for (const AqCfg& c : kAqCfgs) if (c.id == cfg) return c.bk;
  return 0;
}

template <bool W4>
int dispatch_aq(IgemmParams& p, hipStream_t stream, int cfg) {
  switch (cfg) {
#define X(ID, BM, BN, BK, ST, WM, WN, KS, MT) \
  case ID: return launch_tile_aq<BM, BN, BK, ST, WM, WN, W4, KS, MT>(p, stream);
    MIXDQ_AQ_CONFIGS(X)
#undef X
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

}  // namespace
}  // namespace mixdq

using namespace mixdq;

extern "C" int mixdq_qlinear_f16in_select_id(int64_t M, int N, int K, int w4) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 4 != 0 || K % (w4 ? 32 : 16) != 0) return -1;
  const int c = select_aq(M, N, K, w4 != 0);
  if (c > 0 && K % aq_bk(c) == 0) return c;
  return c > 0 && K % 64 == 0 ? 35 : -1;     // K = 320, 960: the 128x128 tile with 64-byte K-tiles
}

// Is one launch cheaper than the reference's two?  Every workgroup quantizes the BM x K activation rows of
// its own tile, so a launch with N / BN column tiles does the quantizer's arithmetic N / BN times over
// (~3.75 vector instructions per element: v_fma_mix, v_rndne, v_cvt_i32, half a v_ashr_pk_i8, a quarter
// v_perm), where the stand-alone kernel does it once, spread over the whole chip, for one launch boundary
// and M * K * 3 bytes of traffic.  Measured on MI355X (tools/bench_f16in.py, profiles/r05_f16in_per_layer.txt,
// microseconds inside a captured graph, two launches -> one): (1024, 1280, 1280) 9.04 -> 9.53,
// (1024, 10240, 1280) 23.7 -> 24.4, (4096, 5120, 640) 28.6 -> 41.8, (76, 1280, 2048) 8.75 -> 10.8 -- and
// (1024, 1280, 640) 7.58 -> 7.23, (16384, 320, 640) 15.9 -> 13.5.  The model below reproduces the sign of
// every row: fused extra ~ 0.5 us + 4e-5 us per element a CU converts; the pair's extra ~ 2.6 us
// (launch + hand-off) + 3 bytes per element at 6 TB/s.
extern "C" int mixdq_qlinear_f16in_preferred(int64_t M, int N, int K, int w4) {
  const int cfg = mixdq_qlinear_f16in_select_id(M, N, K, w4);
  if (cfg <= 0) return 0;
  int bm = 0, bn = 0;
  for (const AqCfg& c : kAqCfgs) if (c.id == cfg) { bm = c.bm; bn = c.bn; }
  const int64_t tiles = ((M + bm - 1) / bm) * (int64_t)((N + bn - 1) / bn);
  const double per_cu = (double)((tiles + kNumCU - 1) / kNumCU) * bm * (double)K;
  const double fused_us = 0.5 + 4e-5 * per_cu;
  const double pair_us = 2.6 + 3.0 * (double)M * (double)K / 6.0e6;
  return fused_us < pair_us ? 1 : 0;
}

extern "C" int mixdq_qlinear_f16in_supported(int64_t M, int N, int K, int64_t lda, int64_t rows, int w4) {
  if (mixdq_qlinear_f16in_select_id(M, N, K, w4) <= 0) return 0;
  if (lda < K || lda % 8 != 0 || rows < M) return 0;
  return (uint64_t)rows * (uint64_t)lda * 2 + 2 * (uint64_t)K < (1ull << 32) &&
         (uint64_t)N * (uint64_t)K < (1ull << 32);
}

extern "C" int mixdq_qlinear_f16in_w8a8(const void* A_f16, int64_t lda, const float* act_scale_inv,
                                        const float* act_zero_point, const int8_t* W,
                                        const float* bias0, const float* scale,
                                        const void* bias_f16_or_null, void* D_f16, int64_t M, int N,
                                        int K, int group_rows, int group_stride, int group_offset,
                                        const void* residual_f16_or_null, int64_t residual_row_div,
                                        int flags, mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || lda < K) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A_f16 || !act_scale_inv || !act_zero_point || !W || !bias0 || !scale || !D_f16)
    return MIXDQ_ERR_INVALID_ARG;
  const bool w4 = flags & MIXDQ_FLAG_W4;
  const bool rowmap = (flags & MIXDQ_FLAG_A_ROWMAP) && group_rows > 0;
  if (N % 4 != 0 || K % (w4 ? 32 : 16) != 0 || lda % 8 != 0) return MIXDQ_ERR_SHAPE;
  if ((((uintptr_t)A_f16 | (uintptr_t)W | (uintptr_t)D_f16 | (uintptr_t)scale | (uintptr_t)bias0 |
        (uintptr_t)residual_f16_or_null) & 15) || ((uintptr_t)bias_f16_or_null & 7))
    return MIXDQ_ERR_SHAPE;
  IgemmParams p{};
  p.A = (const int8_t*)A_f16; p.Wt = W; p.bias0 = bias0; p.scale = scale;
  p.bias = (const __half*)bias_f16_or_null; p.D = (__half*)D_f16;
  p.a_sinv = act_scale_inv; p.a_zp = act_zero_point; p.a_ld = lda; p.a_rowmap = rowmap ? 1 : 0;
  p.M = M; p.N = N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.grp_rows = group_rows; p.grp_stride = group_stride; p.grp_off = group_offset;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  if (p.res && group_rows > 0) return MIXDQ_ERR_ROWMAP_RESIDUAL;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  int cfg = (flags >> 8) & 0xff;
  if (cfg == 0) cfg = mixdq_qlinear_f16in_select_id(M, N, K, w4);
  if (cfg <= 0 || aq_bk(cfg) == 0) return MIXDQ_ERR_SHAPE;
  return w4 ? dispatch_aq<true>(p, (hipStream_t)stream, cfg) : dispatch_aq<false>(p, (hipStream_t)stream, cfg);
}
