// GEMM + residual + LayerNorm + quantize in ONE launch (LNQ instantiations of igemm_kernel, csrc/igemm_kernel.h):
// the residual GEMMs of a transformer block (attn.to_out.0, ff.net.2, proj_in) produce the rows the next
// LayerNorm reads, so the LayerNorm and its consumers' activation quantizers run in the producer's epilogue.
// The reference runs stock nn.LayerNorm followed by its quantize launch (nn/Linear.py:162-164); round 4 ran
// one fused LayerNorm+quantize launch per LayerNorm (210 per step).  Its own translation unit.
#include <cstdlib>
#include "igemm_kernel.h"

namespace mixdq {
namespace {

// exact-fit tiles: 64 x 80 (six stages for K <= 2048, four beyond) and 128 x 80
int select_ln(int64_t M, int N, int K) {
  int units = 1;                                 // the LayerNorm's units per row (oracle/mixdq_oracle.c ln_units)
  while (units < 16 && (N / 16) % (2 * units) == 0) units *= 2;
  if (N % 80 != 0 || N / 80 != units || K % 128 != 0) return -1;   // a column tile = one unit: N = 1280, 640
  // Where it pays (tools/bench_ln_gemm.py, profiles/r05_ln_in_gemm.txt; us per layer, two launches -> one):
  // K <= 1280 -- (4096, 640, 640) 14.5 -> 14.0, (2048, 1280, 1280) 17.3 -> 16.0, (1024, 1280, 1280) 12.8 -> 12.7:
  // the records meet in the XCD's L2 (row blocks laid out XCD by XCD).  Longer K would have every XCD stream
  // all of W through its L2, so its records go through the memory side instead, and that trip costs more than
  // the LayerNorm launch it replaces: (1024, 1280, 5120) 22.1 -> 24.0, (4096, 640, 2560) 18.1 -> 19.4.  Those
  // layers keep their two launches (MIXDQ_LN_MAXK overrides the bound for A/B runs).
  static const int max_k = [] { const char* e = getenv("MIXDQ_LN_MAXK"); return e ? atoi(e) : 1280; }();
  if (K > max_k) return -1;
  const int64_t b64 = ((M + 63) / 64) * (N / 80), b128 = ((M + 127) / 128) * (N / 80);
  if (b64 <= kNumCU) return K > 2048 ? 45 : 56;
  if (b128 <= kNumCU) return 44;
  return -1;
}

constexpr size_t kLnCounterBytes = 4096;      // epoch, departures, sticky error word (three ints; a page of their own)

}  // namespace
}  // namespace mixdq

using namespace mixdq;

extern "C" int mixdq_qlinear_ln_select_id(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  if ((uint64_t)M * (uint64_t)K >= (1ull << 32) || (uint64_t)N * (uint64_t)K >= (1ull << 32)) return -1;
  return select_ln(M, N, K);
}

extern "C" size_t mixdq_qlinear_ln_workspace_bytes(int64_t M, int N) {
  if (M <= 0 || N <= 0) return 0;
  // [0, 4096): the epoch and the departure counter (at a place that does not depend on the problem: a launch
  // of another shape on the same buffer finds them where the last one left them); then one 16-byte record per
  // row and column tile (at most 16 tiles per row)
  return kLnCounterBytes + 2 * (size_t)M * 16 * 16;      // (two copies of the records: csrc/igemm_kernel.h)
}

// The sticky error word of a workspace: 0 while every launch on it found all its records; otherwise the
// tag of the last launch in which a workgroup gave up waiting for one (its rows were written as NaN).  Reads the
// word with a blocking copy on `stream`: for tests and for hosts that check after a synchronisation point, never
// inside a stream capture.  MIXDQ_OK / MIXDQ_ERR_LAUNCH; *status receives the word.
extern "C" int mixdq_qlinear_ln_status(const void* workspace, int* status, mixdq_stream_t stream) {
  if (!workspace || !status) return MIXDQ_ERR_INVALID_ARG;
  int w = 0;
  if (hipMemcpyAsync(&w, (const int*)workspace + 2, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return MIXDQ_ERR_LAUNCH;
  *status = w;
  return MIXDQ_OK;
}

extern "C" int mixdq_qlinear_w8a8_ln(const int8_t* A, const int8_t* W, const float* bias0,
                                     const float* scale, const void* bias_f16_or_null, void* D_f16,
                                     int64_t M, int N, int K, const void* residual_f16_or_null,
                                     int64_t residual_row_div, const void* gamma_f16,
                                     const void* beta_f16, float eps, int n_out,
                                     const float* const* scale_inv, const float* const* zero_point,
                                     int8_t* const* out_q, void* out_f16_or_null, void* workspace,
                                     int flags, mixdq_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || n_out < 0 || n_out > 3) return MIXDQ_ERR_INVALID_ARG;
  if (M == 0 || N == 0) return MIXDQ_OK;
  if (!A || !W || !bias0 || !scale || !D_f16 || !gamma_f16 || !beta_f16 || !workspace ||
      (n_out == 0 && !out_f16_or_null))
    return MIXDQ_ERR_INVALID_ARG;
  if (flags & MIXDQ_FLAG_W4) return MIXDQ_ERR_SHAPE;
  int cfg = (flags >> 8) & 0xff;
  if (cfg == 0) cfg = mixdq_qlinear_ln_select_id(M, N, K);
  if (cfg <= 0) return MIXDQ_ERR_SHAPE;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)D_f16 | (uintptr_t)scale | (uintptr_t)bias0 |
        (uintptr_t)residual_f16_or_null | (uintptr_t)gamma_f16 | (uintptr_t)beta_f16 |
        (uintptr_t)out_f16_or_null | (uintptr_t)workspace) & 15) || ((uintptr_t)bias_f16_or_null & 7))
    return MIXDQ_ERR_ALIGNMENT;
  IgemmParams p{};
  p.A = A; p.Wt = W; p.bias0 = bias0; p.scale = scale; p.bias = (const __half*)bias_f16_or_null;
  p.D = (__half*)D_f16;
  p.M = M; p.N = N; p.Ktot = K;
  p.H = p.W = p.P = p.Q = 1; p.C = K; p.R = p.S = 1; p.stride = 1; p.pad = 0;
  p.res = (const __half*)residual_f16_or_null;
  p.res_div = residual_row_div > 0 ? residual_row_div : 1;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  p.ln_gamma = (const __half*)gamma_f16; p.ln_beta = (const __half*)beta_f16; p.ln_eps = eps;
  p.ln_nq = n_out; p.ln_h = (__half*)out_f16_or_null;
  for (int i = 0; i < n_out; ++i) {
    if (!scale_inv || !zero_point || !out_q || !scale_inv[i] || !zero_point[i] || !out_q[i])
      return MIXDQ_ERR_INVALID_ARG;
    if ((uintptr_t)out_q[i] & 7) return MIXDQ_ERR_ALIGNMENT;
    p.ln_sinv[i] = scale_inv[i]; p.ln_zp[i] = zero_point[i]; p.ln_q[i] = out_q[i];
  }
  // records through the XCD's L2 (row blocks laid out XCD by XCD) where whole row blocks fit an XCD's share of
  // the launch and K is short; MIXDQ_LN_LOCAL=0 / 1 forces either form (A/B runs)
  static const int local_forced = [] { const char* e = getenv("MIXDQ_LN_LOCAL"); return e ? atoi(e) : -1; }();
  p.ln_local = local_forced >= 0 ? (local_forced != 0) : (K <= 2048);
  p.ln_cnt = (int*)workspace;
  p.ln_part = (float*)((char*)workspace + kLnCounterBytes);
  switch (cfg) {
    case 56: return launch_tile_ln<64, 80, 128, 6, 4, 1, 2, 16>(p, (hipStream_t)stream);
    case 45: return launch_tile_ln<64, 80, 128, 4, 4, 1, 2, 16>(p, (hipStream_t)stream);
    case 44: return launch_tile_ln<128, 80, 128, 3, 4, 1, 2, 16>(p, (hipStream_t)stream);
    default: return MIXDQ_ERR_INVALID_ARG;
  }
}

#if MIXDQ_STAMP
// diagnostic builds only (tools/stamp_build.sh): this translation unit's copy of the stamp buffer address
extern "C" int mixdq_debug_stamps_ln(void* buffer) {
  unsigned long long b = (unsigned long long)(uintptr_t)buffer;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &b, sizeof(b)) == hipSuccess ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}
#endif
