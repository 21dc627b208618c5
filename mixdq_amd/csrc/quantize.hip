// a1: FP16 -> INT8 per-tensor affine quantize for gfx950.
// Replaces quantize_kernel.cu:10-48 and quantize_kernel_vectorized.cu:29-95 of the reference
// (one kernel family serves both exported names).
//
// HBM-bound: 2 B read + 1 B written per element.  Dense path: each lane moves 16 B in / 8 B out
// per step, grid-stride, 4 independent loads in flight per lane.  Strided path: dims are sorted
// by input stride and collapsed on the host; the innermost run is vectorised when it is
// contiguous on both sides and 16-B / 8-B aligned.
#include "common.h"

namespace mixdq {
namespace {

struct alignas(16) Half8 { uint32_t w[4]; };
struct alignas(8) Char8 { uint32_t w[2]; };

template <bool UNFUSED>
__device__ __forceinline__ Char8 quantize8(const Half8& h, float s_inv, float zp) {
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t w = h.w[j >> 1];
    __half_raw hr;
    hr.x = (unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu));
    x[j] = __half2float(__half(hr));
  }
  const uint2 q = quantize_pack8<UNFUSED>(x, s_inv, zp);     // common.h: clamp + packing by v_ashr_pk_i8_i32 / v_perm_b32
  Char8 out;
  out.w[0] = q.x;
  out.w[1] = q.y;
  return out;
}

template <bool UNFUSED>
__device__ __forceinline__ int8_t quantize1(const __half* p, float s_inv, float zp) {
  return (int8_t)quantize_one<UNFUSED>(__half2float(*p), s_inv, zp);
}

// Dense: x and out are linear over numel elements.
template <bool UNFUSED>
__global__ __launch_bounds__(256) void quantize_dense_kernel(const __half* __restrict__ x,
                                                             int8_t* __restrict__ out,
                                                             const float* __restrict__ s_inv_p,
                                                             const float* __restrict__ zp_p,
                                                             int64_t numel) {
  const float s_inv = *s_inv_p;
  const float zp = *zp_p;
  const int64_t nvec = numel >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const Half8* xv = reinterpret_cast<const Half8*>(x);
  Char8* ov = reinterpret_cast<Char8*>(out);
  // 4 independent 16-B loads in flight per lane
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    Half8 a = xv[i], b = xv[i + stride], c = xv[i + 2 * stride], d = xv[i + 3 * stride];
    ov[i] = quantize8<UNFUSED>(a, s_inv, zp);
    ov[i + stride] = quantize8<UNFUSED>(b, s_inv, zp);
    ov[i + 2 * stride] = quantize8<UNFUSED>(c, s_inv, zp);
    ov[i + 3 * stride] = quantize8<UNFUSED>(d, s_inv, zp);
  }
  for (; i < nvec; i += stride) ov[i] = quantize8<UNFUSED>(xv[i], s_inv, zp);
  // tail (numel % 8 elements), by the first threads of block 0
  const int64_t tail0 = nvec << 3;
  if (blockIdx.x == 0 && threadIdx.x < (numel - tail0)) {
    out[tail0 + threadIdx.x] = quantize1<UNFUSED>(x + tail0 + threadIdx.x, s_inv, zp);
  }
}

struct StridedArgs {
  int64_t size[4];      // collapsed logical sizes, slowest first; size[3] = inner
  int64_t xs[4];        // x strides (elements)
  int64_t os[4];        // out strides (elements)
};

// Strided, inner dimension contiguous on both sides and vectorisable by 8.
template <bool UNFUSED>
__global__ __launch_bounds__(256) void quantize_rows_kernel(const __half* __restrict__ x,
                                                            int8_t* __restrict__ out,
                                                            const float* __restrict__ s_inv_p,
                                                            const float* __restrict__ zp_p,
                                                            StridedArgs a) {
  const float s_inv = *s_inv_p;
  const float zp = *zp_p;
  const int64_t inner8 = a.size[3] >> 3;
  const int64_t total = a.size[0] * a.size[1] * a.size[2] * inner8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    int64_t v = t % inner8;
    int64_t r = t / inner8;
    int64_t i2 = r % a.size[2];
    r /= a.size[2];
    int64_t i1 = r % a.size[1];
    int64_t i0 = r / a.size[1];
    const __half* xp = x + i0 * a.xs[0] + i1 * a.xs[1] + i2 * a.xs[2] + v * 8;
    int8_t* op = out + i0 * a.os[0] + i1 * a.os[1] + i2 * a.os[2] + v * 8;
    *reinterpret_cast<Char8*>(op) =
        quantize8<UNFUSED>(*reinterpret_cast<const Half8*>(xp), s_inv, zp);
  }
}

// Fully general: one element per thread-step.
template <bool UNFUSED>
__global__ __launch_bounds__(256) void quantize_scalar_kernel(const __half* __restrict__ x,
                                                              int8_t* __restrict__ out,
                                                              const float* __restrict__ s_inv_p,
                                                              const float* __restrict__ zp_p,
                                                              StridedArgs a) {
  const float s_inv = *s_inv_p;
  const float zp = *zp_p;
  const int64_t total = a.size[0] * a.size[1] * a.size[2] * a.size[3];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    int64_t i3 = t % a.size[3];
    int64_t r = t / a.size[3];
    int64_t i2 = r % a.size[2];
    r /= a.size[2];
    int64_t i1 = r % a.size[1];
    int64_t i0 = r / a.size[1];
    out[i0 * a.os[0] + i1 * a.os[1] + i2 * a.os[2] + i3 * a.os[3]] = quantize1<UNFUSED>(
        x + i0 * a.xs[0] + i1 * a.xs[1] + i2 * a.xs[2] + i3 * a.xs[3], s_inv, zp);
  }
}

inline int grid_for(int64_t work_items) {
  int64_t blocks = (work_items + 255) / 256;
  const int64_t cap = (int64_t)kNumCU * 8;   // 2048 blocks, grid-stride the rest
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace
}  // namespace mixdq

using namespace mixdq;

extern "C" int mixdq_quantize_f16_i8(const void* x_f16, int8_t* out, const int64_t* sizes,
                                     const int64_t* x_strides, const int64_t* out_strides,
                                     int ndim, const float* scale_inv, const float* zero_point,
                                     int flags, mixdq_stream_t stream_) {
  if (ndim < 0 || ndim > 8 || !scale_inv || !zero_point) return MIXDQ_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const bool unfused = flags & MIXDQ_FLAG_UNFUSED;
  // ---- sort dims by x stride (descending), drop size-1 dims, collapse mergeable neighbours ----
  int64_t sz[8], xs[8], os[8];
  int n = 0;
  int64_t numel = 1;
  for (int d = 0; d < ndim; ++d) {
    if (sizes[d] < 0 || x_strides[d] < 0 || out_strides[d] < 0) return MIXDQ_ERR_INVALID_ARG;
    numel *= sizes[d];
    if (sizes[d] != 1) {
      sz[n] = sizes[d]; xs[n] = x_strides[d]; os[n] = out_strides[d]; ++n;
    }
  }
  if (numel == 0) return MIXDQ_OK;
  if (!x_f16 || !out) return MIXDQ_ERR_INVALID_ARG;
  for (int i = 1; i < n; ++i)   // insertion sort, stable
    for (int j = i; j > 0 && xs[j - 1] < xs[j]; --j) {
      int64_t t;
      t = sz[j]; sz[j] = sz[j - 1]; sz[j - 1] = t;
      t = xs[j]; xs[j] = xs[j - 1]; xs[j - 1] = t;
      t = os[j]; os[j] = os[j - 1]; os[j - 1] = t;
    }
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (m > 0 && xs[m - 1] == xs[i] * sz[i] && os[m - 1] == os[i] * sz[i]) {
      sz[m - 1] *= sz[i]; xs[m - 1] = xs[i]; os[m - 1] = os[i];
    } else {
      sz[m] = sz[i]; xs[m] = xs[i]; os[m] = os[i]; ++m;
    }
  }
  if (m == 0) { sz[0] = 1; xs[0] = 1; os[0] = 1; m = 1; }
  if (m > 4) return MIXDQ_ERR_SHAPE;
  const __half* x = (const __half*)x_f16;
  const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0);

  if (m == 1 && xs[0] == 1 && os[0] == 1 && aligned) {
    int grid = grid_for((numel >> 3) + 1);
    if (unfused)
      quantize_dense_kernel<true><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, numel);
    else
      quantize_dense_kernel<false><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, numel);
    return launch_status();
  }
  StridedArgs a;
  for (int i = 0; i < 4; ++i) { a.size[i] = 1; a.xs[i] = 0; a.os[i] = 0; }
  for (int i = 0; i < m; ++i) {
    a.size[4 - m + i] = sz[i]; a.xs[4 - m + i] = xs[i]; a.os[4 - m + i] = os[i];
  }
  bool rows_ok = aligned && a.xs[3] == 1 && a.os[3] == 1 && a.size[3] % 8 == 0;
  for (int i = 0; i < 3 && rows_ok; ++i)
    if (a.size[i] > 1 && (a.xs[i] % 8 != 0 || a.os[i] % 8 != 0)) rows_ok = false;
  if (rows_ok) {
    int grid = grid_for(numel >> 3);
    if (unfused)
      quantize_rows_kernel<true><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, a);
    else
      quantize_rows_kernel<false><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, a);
  } else {
    int grid = grid_for(numel);
    if (unfused)
      quantize_scalar_kernel<true><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, a);
    else
      quantize_scalar_kernel<false><<<grid, 256, 0, stream>>>(x, out, scale_inv, zero_point, a);
  }
  return launch_status();
}
