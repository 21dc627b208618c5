// Shared pieces of the FP16 attention core (csrc/attention.hip) that the fused
// to_q + cross-attention epilogue of csrc/igemm.hip reuses: tile geometry, LDS image layout and the
// transposed-read helper.  gfx950 only.
#pragma once
#include "common.h"

namespace mixdq {
namespace {

constexpr int kHeadDim = 64;
constexpr int kKeys = 64;            // keys per tile
constexpr int kRow = 128;            // K and V images: 64 halfs per key, XOR-swizzled 16-B chunks
constexpr int kTileBytes = kKeys * kRow;          // one K (or V) tile
constexpr int kStageBytes = 2 * kTileBytes;       // K tile then V tile
constexpr int kORow = 144;           // output staging row (32 x 144 B per wave)

__device__ __forceinline__ float half_max(float x) {
  // max over lanes l and l^32
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef int v2i __attribute__((ext_vector_type(2)));
union VFrag { struct { v2i lo, hi; } r; v8h h; };

// Two transposed reads (keys k..k+3 and k+8..k+11 of 16 output columns) = one A operand.
template <int OFF>
__device__ __forceinline__ void tr_read2_imm(VFrag& f, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(f.r.lo), "=&v"(f.r.hi)
               : "v"(addr), "n"(OFF), "n"(OFF + 8 * kRow)
               : "memory");
}
__device__ __forceinline__ void s_waitcnt_lgkm0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep register-only MFMAs behind the wait
}

}  // namespace
}  // namespace mixdq
