// FP16 attention core for gfx950 (head_dim 64): O = softmax(Q K^T * scale) V, flash-style.
//
// The reference keeps the attention matmuls in FP16 in both of its paths (SURVEY.md §0:
// quant_block.py:630-637 — plain get_attention_scores + torch.bmm; only to_q/to_k/to_v/to_out.0 are
// quantized).  This kernel is that FP16 core, written for CDNA4, reading q/k/v straight out of the
// (fused) projection output — head h is the 64-column slice h*64.. of each row, any row stride —
// and writing either FP16 rows or, fused, the INT8 operand of to_out.0 (same quantize arithmetic as
// mixdq_quantize_f16_i8 applied to the FP16-rounded output).
//
// Structure per workgroup: WAVES waves x 32 query rows; K/V tiles of 64 keys arrive by LDS-DMA into a
// ring of four (three in flight, one barrier per tile).  Each wave runs a three-deep software
// pipeline over the tiles (attn_tile_step): the softmax of tile t on the VALU next to the MFMAs of
// P V (t-1) and Q K^T (t+1), interleaved by hand at two waves per SIMD (<= 256 registers).
//   S^T = K Q^T      v_mfma_f32_32x32x16_f16, A = K rows (ds_read_b128, XOR-swizzled image),
//                    B = Q^T held in registers.  Lane (q = lane%32, h = lane/32) then owns, for ITS
//                    query, keys 32kb + 8g + 4h + 0..3: the row max / row sum are in-lane reductions
//                    plus one v_permlane32_swap.
//   O^T = V^T P^T    B = P^T is the S^T accumulator itself, converted to FP16 in place (k-slot j of
//                    half h <-> key 32kb + 16u + 8(j/4) + 4h + j%4); A = V^T comes from the row-major
//                    V image through ds_read_b64_tr_b16 with the same slot order (128-B rows, 16-B
//                    chunks XOR-swizzled: the four rows of a transposed read fall in four disjoint
//                    16-bank ranges).
// Softmax in FP32 with base-2 exponentials (v_exp_f32), P rounded to FP16 for the second MFMA, row
// sums accumulated in FP32 from that rounded P (a third MFMA with a ones operand); O staged through LDS and stored as whole 128-B rows.
#include "common.h"
#include <cstdlib>
#include "attn_core.h"

namespace mixdq {
namespace {

typedef short v4s16 __attribute__((__vector_size__(4 * sizeof(short))));

// MIXDQ_STAMP (diagnostic builds only, tools/stamp_build.sh): every wave of every attention workgroup records the
// shader clock at the phase boundaries of attn_fwd_kernel into a buffer registered with mixdq_debug_stamps_attn();
// tools/stamp_report.py --attn turns them into a time line (where the 1024-token launch spends the
// ~7 us outside its 16-tile loop).
#ifndef MIXDQ_STAMP
#define MIXDQ_STAMP 0
#endif
#if MIXDQ_STAMP
__device__ unsigned long long g_attn_stamps;      // address of [workgroup][wave 0..3][16] uint64, or 0
#define MIXDQ_ATTN_STAMP(slot)                                                                    \
  do {                                                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                   \
    const unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                               \
    const unsigned long long a_ = g_attn_stamps;                                                  \
    if (a_ != 0 && (threadIdx.x & 63) == 0) {                                                     \
      auto sp_ = (__attribute__((address_space(1))) unsigned long long*)a_ +                      \
                 ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;                              \
      sp_[slot] = t_;                                                                             \
      if ((slot) == 0) sp_[8] = r_;                                                               \
      if ((slot) == 7) sp_[9] = r_;                                                               \
    }                                                                                             \
  } while (0)
#else
#define MIXDQ_ATTN_STAMP(slot) do {} while (0)
#endif

struct AttnParams {
  const __half* q; const __half* k; const __half* v;
  void* out;                       // f16 rows or int8 rows
  long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;   // batch / row strides in elements
  int tq, tkv, heads, qblocks;
  int xcd_map;                     // 1: workgroup -> (batch, head, query block) through attn_block_of()
  float scale_log2;                // softmax scale * log2(e)
  const float* s_inv; const float* zp;
  int unfused;
  // prefetch payload (mixdq_attention_f16_prefetch): workgroups attn_blocks .. attn_blocks + pf_blocks - 1
  // read these byte ranges and do nothing else
  int attn_blocks, pf_blocks, n_pf, pf_nt;
  int pf_delay;                    // payload workgroups sleep pf_delay x 127 x 64 clocks (~3.9 us each) before their first load
  const char* pf_ptr[16];
  long pf_bytes[16];
};

// The prefetch role: every thread reads 16 bytes per step, a workgroup 4 KB, the payload workgroups
// together a contiguous stripe; eight loads in flight per lane.  The values are folded into a register the
// compiler must keep (an empty asm consumes it): nothing is written.
__device__ __forceinline__ void attn_prefetch_role(const AttnParams& p) {
  // The payload starts LATE: its workgroups are dispatched within a microsecond of the attention's, and 24 MB of
  // loads in flight at that moment sit in front of the attention's own first requests (Q, the first K/V tiles:
  // entry -> prologue issued 2.4 -> 5.6 us, tools/stamp_attn.py --payload 24); the attention needs the memory
  // side for ~4 us and then runs out of L2 / LDS, which is when the payload should stream.
  for (int i = 0; i < p.pf_delay; ++i) __builtin_amdgcn_s_sleep(127);
  const long first = ((long)(blockIdx.x - p.attn_blocks) * blockDim.x + threadIdx.x) * 16;
  const long stride = (long)p.pf_blocks * blockDim.x * 16;
  v4i acc = {0, 0, 0, 0};
  for (int r = 0; r < p.n_pf; ++r) {
    const char* base = p.pf_ptr[r];
    const long n = p.pf_bytes[r] & ~15l;           // whole 16-byte pieces (the tail shares their cache line)
    long off = first;
    for (; off + 7 * stride < n; off += 8 * stride) {
      v4i t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const v4i* src = reinterpret_cast<const v4i*>(base + off + u * stride);
        t[u] = p.pf_nt ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc ^= t[u];
    }
    for (; off < n; off += stride) {
      const v4i* src = reinterpret_cast<const v4i*>(base + off);
      acc ^= p.pf_nt ? __builtin_nontemporal_load(src) : *src;
    }
  }
  asm volatile("" ::"v"(acc));
}

// XCD-aware workgroup map (round 6).  Workgroups are dealt round-robin over the 8 XCDs (bid % 8), each with its own
// 4 MiB L2, and all query blocks of one (batch, head) read the SAME K / V.  With block = bid the query blocks of a
// head were spread over all 8 XCDs and every L2 fetched every head's K / V: 4.4x the algorithmic bytes on the
// memory side at every shape (profiles/pmc_traffic.json, round 5: 8 x (K + V) + Q + O).  Here XCD x -- blocks
// x, x + 8, ... -- gets a CONTIGUOUS run of the (batch, head, query block) sequence (bijective for any count, as
// igemm's tile map), so a head's query blocks meet in one L2 (a head that straddles two runs: two).  Which XCD a
// block lands on is an observation, not a promise: the map changes where a block runs, never what it computes.
__device__ __forceinline__ int attn_block_of(int bid, int total) {
  const int x = bid % kNumXCD, q8 = total / kNumXCD, r8 = total % kNumXCD;
  return (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + bid / kNumXCD;
}

__device__ __forceinline__ float half_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void lds_read128_imm(v8h& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}

// Running state of one wave's 32 query rows.
struct AttnState {
  v16f o[2], lsum;               // O^T accumulators (d 0..31, 32..63); row sums, every register equal
  float m_i;                     // running row maximum
  VFrag vf[2][2][2];             // V^T fragments of the tile whose P V product is issued next
};

__device__ __forceinline__ void attn_state_init(AttnState& st) {
#pragma unroll
  for (int i = 0; i < 16; ++i) { st.o[0][i] = 0.f; st.o[1][i] = 0.f; st.lsum[i] = 0.f; }
  st.m_i = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {                  // "tile -1": a zero product
    st.vf[i >> 2][(i >> 1) & 1][i & 1].r.lo = v2i{0, 0};
    st.vf[i >> 2][(i >> 1) & 1][i & 1].r.hi = v2i{0, 0};
  }
}

// O^T += V^T P^T and the row sums for one whole tile (the pipeline's drain).
__device__ __forceinline__ void attn_pv_tile(AttnState& st, const v8h (&pu)[2][2], const v8h& ones) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int db = 0; db < 2; ++db)
        st.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(st.vf[kb][u][db].h, pu[kb][u], st.o[db], 0, 0, 0);
      st.lsum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pu[kb][u], st.lsum, 0, 0, 0);
    }
}

// One iteration of the software pipeline, for the wave's tile t ("this"), the one before it
// ("last") and the one after it ("next").  A wave issues in order and an MFMA runs for 32 cycles
// after its issue, so matrix work hides under VALU work only when the two are interleaved
// instruction by instruction and independent.  The iteration therefore pairs the softmax of this
// tile (VALU: maximum, 32 exponentials, FP16 conversion; scores `sc` -> probabilities `pc`) with
// MFMAs that do not depend on it: P V of the last tile (12, row sums included; `pl` and st.vf) and
// Q K^T of the next one (8, into `sn`).  hipcc's own placement clusters the MFMAs in front of the
// VALU block (sched_group_barrier requests notwithstanding), so the order is pinned by hand.
//   k_next[ks]: LDS address of this lane's K row chunk of k-step ks in the next tile's image
//   v0 / v1   : LDS addresses of this lane's transposed V reads (d 0..31 / 32..63) in this tile's
// Every LDS read is inline asm and counted by hand: hipcc would put `s_waitcnt vmcnt(0)` in front
// of a ds_read_tr builtin while an LDS-DMA is in flight (draining the prefetch ring every tile) and
// cannot count asm reads next to its own.  On return st.vf holds this tile's V^T fragments, still
// in flight: the caller waits lgkmcnt(0) before the next step (or the drain).
template <bool RAGGED>
__device__ __forceinline__ void attn_tile_step(AttnState& st, v16f (&sc)[2], v16f (&sn)[2],
                                               v8h (&pc)[2][2], v8h (&pl)[2][2],
                                               const v8h (&qf)[4], const v8h& ones, float c,
                                               const unsigned (&k_next)[4], unsigned v0, unsigned v1,
                                               bool mask_next, int lim_next) {
  v16f (&o)[2] = st.o;
  v16f& lsum = st.lsum;
  VFrag (&vf)[2][2][2] = st.vf;
  v8h kf[2][2];                                  // two k-steps at a time: registers
  auto pv = [&](int g) {                         // keys 16 g .. 16 g + 15 of the last tile
    const int kb = g >> 1, u = g & 1;
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][u][0].h, pl[kb][u], o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kb][u][1].h, pl[kb][u], o[1], 0, 0, 0);
    lsum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pl[kb][u], lsum, 0, 0, 0);
  };
  auto qk1 = [&](int i) {                        // one k-step of the next tile's scores
    const int kb = i & 1, ks = i >> 1;
    const v16f zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks & 1], qf[ks], ks ? sn[kb] : zero, 0, 0, 0);
  };
  auto k_read = [&](int ks0) {                   // K fragments of k-steps ks0, ks0 + 1
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      lds_read128_imm<0>(kf[0][ks], k_next[ks0 + ks]);
      lds_read128_imm<32 * kRow>(kf[1][ks], k_next[ks0 + ks]);
    }
  };
  float mc = 0.f;
  auto ex = [&](int i) {                         // four exponentials -> two packed FP16 pairs
    const int kb = i >> 2;
#pragma unroll
    for (int r = 4 * (i & 3); r < 4 * (i & 3) + 4; ++r)
      pc[kb][r >> 3][r & 7] = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], c, -mc));
  };
  // A stage = a few MFMAs + a slice of the softmax, fenced by empty asm statements that the
  // stage's inputs and results pass through ("+v"): volatile asms keep their order, so nothing
  // of a stage can be hoisted above its opening fence or sink below its closing one.
#define MIXDQ_FENCE_PV(B)  asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(lsum), "+v"(B), "+v"(mc))
#define MIXDQ_FENCE_QK(A0, A1, PCV) \
  asm volatile("" : "+v"(sn[0]), "+v"(sn[1]), "+v"(A0), "+v"(A1), "+v"(mc), "+v"(PCV))

  // ---- maximum of this tile (lane = one query row; the other 32 keys: lane ^ 32) ----
  MIXDQ_FENCE_PV(pl[0][0]);
  pv(0);
  float mx = sc[0][0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
  asm volatile("" : "+v"(mx));
  MIXDQ_FENCE_PV(pl[0][1]);
  pv(1);
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
  mx = half_max(mx);
  const float m_new = fmaxf(st.m_i, mx);
  const bool grew = m_new > st.m_i;
  mc = m_new * c;
  // ---- exponentials, four keys at a time behind one or two MFMAs ----
  MIXDQ_FENCE_PV(pl[1][0]);
  k_read(0);                                     // in flight until the first Q K^T stage
  pv(2); ex(0);
  asm volatile("" : "+v"(pc[0][0]));
  MIXDQ_FENCE_PV(pl[1][1]);
  pv(3); ex(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the K fragments
  asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(lsum));
  asm volatile("" : "+v"(kf[0][0]), "+v"(kf[1][0]), "+v"(mc), "+v"(pc[0][0]));
  qk1(0); qk1(1); ex(2);
  MIXDQ_FENCE_QK(kf[0][1], kf[1][1], pc[0][1]);
  qk1(2); qk1(3); ex(3);
  MIXDQ_FENCE_QK(kf[0][0], kf[1][0], pc[0][1]);
  k_read(2);
  ex(4);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  MIXDQ_FENCE_QK(kf[0][0], kf[1][0], pc[1][0]);
  // V^T fragments of this tile: the products of the last one were issued stages ago
  tr_read2_imm<0 * kRow>(vf[0][0][0], v0);
  tr_read2_imm<0 * kRow>(vf[0][0][1], v1);
  tr_read2_imm<16 * kRow>(vf[0][1][0], v0);
  tr_read2_imm<16 * kRow>(vf[0][1][1], v1);
  tr_read2_imm<32 * kRow>(vf[1][0][0], v0);
  tr_read2_imm<32 * kRow>(vf[1][0][1], v1);
  tr_read2_imm<48 * kRow>(vf[1][1][0], v0);
  tr_read2_imm<48 * kRow>(vf[1][1][1], v1);
  qk1(4); qk1(5); ex(5);
  MIXDQ_FENCE_QK(kf[0][1], kf[1][1], pc[1][0]);
  qk1(6); ex(6);
  asm volatile("" : "+v"(sn[0]), "+v"(kf[1][1]), "+v"(mc), "+v"(pc[1][1]));
  qk1(7); ex(7);
  asm volatile("" : "+v"(sn[1]), "+v"(pc[1][1]));
#undef MIXDQ_FENCE_PV
#undef MIXDQ_FENCE_QK
  if (RAGGED && mask_next) {                     // mask the absent keys of the last tile
    asm volatile("" ::: "memory");               // a real branch: not worth if-converting
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * kb + 8 * (r >> 2) + (r & 3) >= lim_next) sn[kb][r] = -INFINITY;
  }
  if (__builtin_amdgcn_ballot_w64(grew)) {       // some row's maximum moved: rescale O and sums
    const float alpha = __builtin_amdgcn_exp2f((st.m_i - m_new) * c);
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    lsum[0] *= alpha;
  }
  st.m_i = m_new;
}

template <int WAVES, int STAGES, bool QUANT, bool RAGGED>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_fwd_kernel(const AttnParams p) {
  MIXDQ_ARGS_NOW(p.q, p.k, p.v, p.out, p.q_bs, p.q_rs, p.k_bs, p.k_rs, p.v_bs, p.v_rs, p.o_bs, p.o_rs,
                 p.tq, p.tkv, p.heads, p.qblocks);
  MIXDQ_ARGS_NOW(p.scale_log2, p.s_inv, p.zp, p.unfused, p.xcd_map, p.attn_blocks);
  if ((int)blockIdx.x >= p.attn_blocks) {        // payload workgroups (dispatched after the attention ones)
    attn_prefetch_role(p);
    return;
  }
  MIXDQ_ATTN_STAMP(0);
  // the output quantizer's scalars: requested at entry, as SCALAR loads (constant address space: they do not
  // change while the kernel runs).  Read where they are used -- behind the last barrier -- they were a
  // dependent trip to memory at the very end of every launch: stores phase 1.5 us (tools/stamp_attn.py).
  float s_inv = 0.f, zp = 0.f;
  if constexpr (QUANT) {
    s_inv = *(const __attribute__((address_space(4))) float*)p.s_inv;
    zp = *(const __attribute__((address_space(4))) float*)p.zp;
  }
  constexpr int NI = 16 / WAVES;                 // LDS-DMA wave-instructions per wave per tile
  constexpr int PRE = STAGES - 1;                // tiles staged ahead of the one whose V is consumed
  static_assert(STAGES >= 3, "tiles t (V) and t+1 (K) are read while t+2.. are in flight");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keep it scalar
  const int l32 = lane & 31, hh = lane >> 5;
  const int blk = p.xcd_map ? attn_block_of(blockIdx.x, p.attn_blocks) : (int)blockIdx.x;
  const int qb = blk % p.qblocks;
  const int head = (blk / p.qblocks) % p.heads;
  const int b = blk / (p.qblocks * p.heads);
  const int q0 = qb * (WAVES * 32) + wave * 32;

  // Q^T fragments: lane's query row, d = 16 ks + 8 h .. + 7
  v8h qf[4];
  {
    const int qr = min(q0 + l32, p.tq - 1);
    const __half* qrow = p.q + b * p.q_bs + (long)qr * p.q_rs + head * kHeadDim + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const v8h*>(qrow + ks * 16);
  }

  // LDS-DMA staging: one wave-instruction fills 8 keys x 128 B, lane -> (key l/8, slot l%8); the
  // slot holds chunk slot ^ swizzle(key), so the SOURCE address carries the permutation.
  //   K image: chunk ^ ((key >> 1) & 7)     (conflict-free ds_read_b128 of the A operand)
  //   V image: chunk ^ 4*((key >> 1) & 1)   (the 4 rows of a transposed read -> 4 disjoint ranges)
  // A wave's NI instructions are all-K or all-V (waves 0..WAVES/2-1 stage K, the rest V).
  const int srow = lane >> 3, spos = lane & 7;
  const bool stage_v = wave * NI >= 8;
  // wave-uniform byte base (SGPR pair) + 32-bit per-lane byte offsets: the saddr form of the DMA
  const char* sbase = reinterpret_cast<const char*>(
      (stage_v ? p.v + b * p.v_bs : p.k + b * p.k_bs) + head * kHeadDim);
  const unsigned srs = 2u * (unsigned)(stage_v ? p.v_rs : p.k_rs);   // key row stride in bytes
  const int last_key = p.tkv - 1;
  unsigned row_off[NI], col_off[NI];             // byte offsets of this lane's rows / chunks
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int krow = ((wave * NI + i) & 7) * 8 + srow;     // key row within the tile
    const int sw = stage_v ? ((krow >> 1) & 1) << 2 : (krow >> 1) & 7;
    col_off[i] = (spos ^ sw) * 16;
    row_off[i] = krow * srs + col_off[i];
  }
  auto stage = [&](int buf, int t) {
    char* dst = smem + buf * kStageBytes + wave * NI * 1024;
    if ((t + 1) * kKeys <= p.tkv) {                       // whole tile in range (wave-uniform)
      const char* tile = sbase + (size_t)((unsigned)(t * kKeys) * srs);
#pragma unroll
      for (int i = 0; i < NI; ++i) glds16(tile + row_off[i], dst + i * 1024);
    } else {
      // keys past the end re-read the last key: finite data whose scores are masked to -inf below
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int key = min(t * kKeys + ((wave * NI + i) & 7) * 8 + srow, last_key);
        glds16(sbase + ((unsigned)key * srs + col_off[i]), dst + i * 1024);
      }
    }
  };

  // per-lane LDS read offsets
  int k_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd[ks] = l32 * kRow + (((2 * ks + hh) ^ ((l32 >> 1) & 7)) << 4);
  const int q4 = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
  const int v_rd0 = kTileBytes + (4 * hh + q4) * kRow +
                    (((2 * g16 + (pp >> 1)) ^ (((q4 >> 1) & 1) << 2)) << 4) + 8 * (pp & 1);
  const int v_rd1 = v_rd0 ^ 64;                  // output columns 32..63: chunk index ^ 4
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned v_a0 = lds0 + v_rd0, v_a1 = lds0 + v_rd1;
  unsigned k_a[4];                               // absolute LDS addresses: offsets fold to immediates
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_a[ks] = lds0 + k_rd[ks];

  const int ntiles = (p.tkv + kKeys - 1) / kKeys;

  // S^T = K Q^T for the tile in buffer `buf` (tile index t for the ragged-tail mask)
  auto qk = [&](int buf, int t, v16f (&s)[2]) {
    v8h kf[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        kf[kb][ks] = *(const __attribute__((address_space(3))) v8h*)(size_t)(
            k_a[ks] + (buf * kStageBytes + kb * 32 * kRow));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb][ks], qf[ks], s[kb], 0, 0, 0);
    }
    if (RAGGED && t == ntiles - 1) {               // mask the absent keys of the last tile
      asm volatile("" ::: "memory");               // a real branch: not worth if-converting
      const int lim = p.tkv - t * kKeys - 4 * hh;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * kb + 8 * (r >> 2) + (r & 3) >= lim) s[kb][r] = -INFINITY;
    }
  };

  static_assert(STAGES % 2 == 0, "the score registers ping-pong with the buffer parity");
  AttnState st;
  attn_state_init(st);
  const float c = p.scale_log2;
  v8h ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (_Float16)1.f;

  // SHORT key sequences (every tile fits the prologue: cross-attention, 77 keys = 2 tiles): only the
  // real tiles are staged, all of them up front, and the loop below stages nothing -- its counted
  // waits are then satisfied at once.  (Staged like a long sequence, 3 of the 5 stages a workgroup
  // issued were dummies re-reading the last key: 48 of 80 DMA pieces.)  Same arithmetic, same order.
  const bool short_k = ntiles <= PRE;            // wave-uniform
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (!short_k || s < ntiles) stage(s, s);     // long: tiles past the end re-stage the last key
  MIXDQ_ATTN_STAMP(1);                           // Q requested, the prologue's K/V tiles requested
  if (short_k) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((PRE - 1) * NI) : "memory");
  MIXDQ_ATTN_STAMP(2);                           // the first K/V tile has landed (every wave's pieces)
  v16f S[2][2];                                  // scores: S[t & 1] softmaxed now, S[~t & 1] next
  qk(0, 0, S[0]);
  MIXDQ_ATTN_STAMP(3);                           // Q in registers, scores of tile 0
  v8h P[2][2][2];                                // P[t & 1]: probabilities of tile t, FP16, B operand
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) P[1][kb][u][j] = (_Float16)0.f;

  // Software pipeline (attn_tile_step): iteration t = softmax(t) next to P V(t-1) and Q K^T(t+1).
  for (int t0 = 0; t0 < ntiles; t0 += STAGES) {
#pragma unroll
    for (int sb = 0; sb < STAGES; ++sb) {          // tile t0 + sb lives in buffer sb (compile-time)
      const int t = t0 + sb;
      if (t >= ntiles) break;
      // tile t+1 has landed (PRE-2 younger tiles may still fly); every wave is past tile t-1 --
      // its V fragments are in registers (lgkmcnt) -- whose buffer is restaged next
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((PRE - 2) * NI) : "memory");
      if (!short_k) stage((sb + PRE) % STAGES, t + PRE);
      unsigned k_next[4];                          // past the last tile: a re-staged key, unused
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) k_next[ks] = k_a[ks] + ((sb + 1) % STAGES) * kStageBytes;
      attn_tile_step<RAGGED>(st, S[sb & 1], S[(sb + 1) & 1], P[sb & 1], P[(sb + 1) & 1], qf, ones, c,
                             k_next, v_a0 + sb * kStageBytes, v_a1 + sb * kStageBytes,
                             t + 1 == ntiles - 1, p.tkv - (t + 1) * kKeys - 4 * hh);
    }
  }
  MIXDQ_ATTN_STAMP(4);                           // the tile loop is done
  // ---- the last tile's product ----
  s_waitcnt_lgkm0();
  if ((ntiles - 1) & 1) attn_pv_tile(st, P[1], ones); else attn_pv_tile(st, P[0], ones);
  v16f (&o)[2] = st.o;
  v16f& lsum = st.lsum;
  // the DMAs staged for tiles >= ntiles may still be in flight: drain before LDS reuse
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- normalise, stage through LDS, store whole rows ----
  const float inv = 1.f / lsum[0];
  char* Os = smem + wave * (32 * kORow);
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      v4h w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (_Float16)(o[db][4 * g + j] * inv);
      *reinterpret_cast<v4h*>(Os + l32 * kORow + (32 * db + 8 * g + 4 * hh) * 2) = w;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging (a wave reads back its own 32 rows): no block barrier
  MIXDQ_ATTN_STAMP(5);                           // O normalised and staged in LDS
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = lane + 64 * i, row = id >> 3, ch = id & 7;
    if (q0 + row >= p.tq) continue;
    const uint4 w = *reinterpret_cast<const uint4*>(Os + row * kORow + ch * 16);
    const long off = b * p.o_bs + (long)(q0 + row) * p.o_rs + head * kHeadDim + ch * 8;
    if constexpr (!QUANT) {
      *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(p.out) + off) = w;
    } else {
      const __half* hv = reinterpret_cast<const __half*>(&w);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __half2float(hv[j]);
      *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(p.out) + off) =
          p.unfused ? quantize_pack8<true>(x, s_inv, zp) : quantize_pack8<false>(x, s_inv, zp);
    }
  }
  MIXDQ_ATTN_STAMP(7);                           // rows stored (issued)
}

// SHORT key sequences (tkv <= 128: the UNet's cross-attention, 77 keys): a workgroup's life is one
// memory round trip (K / V staged, Q fetched) and ~1.5 us of arithmetic -- a latency chain, and the
// pipelined kernel above holds 2 waves per SIMD (its three-deep pipeline needs 200+ registers), so at
// batch >= 2, where the fused to_q + cross-attention launch (igemm.hip) is not used, the launch ran in
// 2.5 rounds of two workgroups per CU.  This kernel runs the same arithmetic in the same order --
// tile by tile, unpipelined, exactly the sequence of the fused epilogue in igemm.hip -- in <= 128
// registers: four workgroups per CU, 32 KB of LDS each.  Bit-identical to attn_fwd_kernel
// (tests/test_attention_gpu.py).
template <bool QUANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_short_kernel(const AttnParams p) {
  MIXDQ_ARGS_NOW(p.q, p.k, p.v, p.out, p.q_bs, p.q_rs, p.k_bs, p.k_rs, p.v_bs, p.v_rs, p.o_bs, p.o_rs,
                 p.tq, p.tkv, p.heads, p.qblocks);
  MIXDQ_ARGS_NOW(p.scale_log2, p.s_inv, p.zp, p.unfused, p.xcd_map, p.attn_blocks);
  float s_inv = 0.f, zp = 0.f;                   // (scalar loads at entry: see attn_fwd_kernel)
  if constexpr (QUANT) {
    s_inv = *(const __attribute__((address_space(4))) float*)p.s_inv;
    zp = *(const __attribute__((address_space(4))) float*)p.zp;
  }
  constexpr int WAVES = 4, NI = 16 / WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, hh = lane >> 5;
  const int blk = p.xcd_map ? attn_block_of(blockIdx.x, p.attn_blocks) : (int)blockIdx.x;
  const int qb = blk % p.qblocks;
  const int head = (blk / p.qblocks) % p.heads;
  const int b = blk / (p.qblocks * p.heads);
  const int q0 = qb * (WAVES * 32) + wave * 32;
  const int ntiles = (p.tkv + kKeys - 1) / kKeys;      // 1 or 2

  v8h qf[4];
  {
    const int qr = min(q0 + l32, p.tq - 1);
    const __half* qrow = p.q + b * p.q_bs + (long)qr * p.q_rs + head * kHeadDim + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const v8h*>(qrow + ks * 16);
  }
  // staging: as attn_fwd_kernel (waves 0, 1: K, waves 2, 3: V; same images), real tiles only
  {
    const int srow = lane >> 3, spos = lane & 7;
    const bool stage_v = wave * NI >= 8;
    const char* sbase = reinterpret_cast<const char*>(
        (stage_v ? p.v + b * p.v_bs : p.k + b * p.k_bs) + head * kHeadDim);
    const unsigned srs = 2u * (unsigned)(stage_v ? p.v_rs : p.k_rs);
    const int last_key = p.tkv - 1;
    for (int t = 0; t < ntiles; ++t) {
      char* dst = smem + t * kStageBytes + wave * NI * 1024;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int krow = ((wave * NI + i) & 7) * 8 + srow;
        const int sw = stage_v ? ((krow >> 1) & 1) << 2 : (krow >> 1) & 7;
        const int key = min(t * kKeys + krow, last_key);   // past the end: the last key, masked below
        glds16(sbase + ((unsigned)key * srs + (unsigned)((spos ^ sw) * 16)), dst + i * 1024);
      }
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned k_a[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_a[ks] = lds0 + l32 * kRow + (((2 * ks + hh) ^ ((l32 >> 1) & 7)) << 4);
  const int q4 = (lane & 15) >> 2, pp = lane & 3, g16 = (lane >> 4) & 1;
  const int v_rd0 = kTileBytes + (4 * hh + q4) * kRow +
                    (((2 * g16 + (pp >> 1)) ^ (((q4 >> 1) & 1) << 2)) << 4) + 8 * (pp & 1);
  const unsigned v_a0 = lds0 + v_rd0, v_a1 = lds0 + (v_rd0 ^ 64);
  v16f o[2], lsum;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; lsum[i] = 0.f; }
  float m_i = -INFINITY;
  const float c = p.scale_log2;
  v8h ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (_Float16)1.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t >= ntiles) break;
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
    if (t == ntiles - 1 && (p.tkv & (kKeys - 1)) != 0) {     // mask the absent keys
      const int lim = p.tkv - t * kKeys - 4 * hh;
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
        pf[kb][r >> 3][r & 7] = (_Float16)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[kb][r], c, -mc));
    if (__builtin_amdgcn_ballot_w64(grew)) {
      const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c);
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
      lsum[0] *= alpha;
    }
    m_i = m_new;
    // V^T fragments two 16-key groups at a time (registers), in the product order of attn_pv_tile
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      VFrag vf[2][2];
      const unsigned a0 = v_a0 + t * kStageBytes, a1 = v_a1 + t * kStageBytes;
      if (kb == 0) {
        tr_read2_imm<0 * kRow>(vf[0][0], a0);
        tr_read2_imm<0 * kRow>(vf[0][1], a1);
        tr_read2_imm<16 * kRow>(vf[1][0], a0);
        tr_read2_imm<16 * kRow>(vf[1][1], a1);
      } else {
        tr_read2_imm<32 * kRow>(vf[0][0], a0);
        tr_read2_imm<32 * kRow>(vf[0][1], a1);
        tr_read2_imm<48 * kRow>(vf[1][0], a0);
        tr_read2_imm<48 * kRow>(vf[1][1], a1);
      }
      s_waitcnt_lgkm0();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[u][db].h, pf[kb][u], o[db], 0, 0, 0);
        lsum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf[kb][u], lsum, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                   // every wave is done with the K / V images
  const float inv = 1.f / lsum[0];
  char* Os = smem + wave * (32 * kORow);
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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = lane + 64 * i, row = id >> 3, ch = id & 7;
    if (q0 + row >= p.tq) continue;
    const uint4 w = *reinterpret_cast<const uint4*>(Os + row * kORow + ch * 16);
    const long off = b * p.o_bs + (long)(q0 + row) * p.o_rs + head * kHeadDim + ch * 8;
    if constexpr (!QUANT) {
      *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(p.out) + off) = w;
    } else {
      const __half* hv = reinterpret_cast<const __half*>(&w);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __half2float(hv[j]);
      *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(p.out) + off) =
          p.unfused ? quantize_pack8<true>(x, s_inv, zp) : quantize_pack8<false>(x, s_inv, zp);
    }
  }
}

int launch_attn_short(AttnParams& p, int batch, bool quant, hipStream_t stream) {
  if ((long)p.qblocks * p.heads * batch > 0x7fffffffl) return MIXDQ_ERR_INVALID_ARG;
  const int grid = p.qblocks * p.heads * batch;
  p.attn_blocks = grid;
  const int smem = 2 * kStageBytes;                  // two K | V tiles; the output staging overlays them
  if (quant) hipLaunchKernelGGL((attn_short_kernel<true>), dim3(grid), dim3(256), smem, stream, p);
  else hipLaunchKernelGGL((attn_short_kernel<false>), dim3(grid), dim3(256), smem, stream, p);
  return launch_status();
}

constexpr int attn_smem_bytes(int waves, int stages) {
  const int kv = stages * kStageBytes;
  const int os = waves * 32 * kORow;
  return kv > os ? kv : os;
}

template <int WAVES, int STAGES>
int launch_attn(AttnParams& p, int batch, bool quant, hipStream_t stream) {
  p.attn_blocks = p.qblocks * p.heads * batch;
  if (p.n_pf == 0) p.pf_blocks = 0;
  if ((long)p.attn_blocks + p.pf_blocks > 0x7fffffffl) return MIXDQ_ERR_INVALID_ARG;
  const int grid = p.attn_blocks + p.pf_blocks;
  const int smem = attn_smem_bytes(WAVES, STAGES);
  const dim3 g(grid), b(WAVES * 64);
  const bool ragged = (p.tkv & (kKeys - 1)) != 0;   // whole key tiles: no masking code at all
#define MIXDQ_ATTN_LAUNCH(Q, R) \
  hipLaunchKernelGGL((attn_fwd_kernel<WAVES, STAGES, Q, R>), g, b, smem, stream, p)
  if (quant) { if (ragged) MIXDQ_ATTN_LAUNCH(true, true); else MIXDQ_ATTN_LAUNCH(true, false); }
  else       { if (ragged) MIXDQ_ATTN_LAUNCH(false, true); else MIXDQ_ATTN_LAUNCH(false, false); }
#undef MIXDQ_ATTN_LAUNCH
  return launch_status();
}

}  // namespace
}  // namespace mixdq

using namespace mixdq;

static int attention_f16_impl(const void* q, const void* k, const void* v, void* out,
                              int batch, int heads, int head_dim, int tq, int tkv,
                              int64_t q_batch_stride, int64_t q_row_stride,
                              int64_t k_batch_stride, int64_t k_row_stride,
                              int64_t v_batch_stride, int64_t v_row_stride,
                              int64_t out_batch_stride, int64_t out_row_stride,
                              float softmax_scale, const float* out_scale_inv,
                              const float* out_zero_point, const void* const* pf_ptrs,
                              const int64_t* pf_bytes, int n_pf, int flags, mixdq_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_pf < 0 || n_pf > 16 || (n_pf > 0 && (!pf_ptrs || !pf_bytes))) return MIXDQ_ERR_INVALID_ARG;
  if (batch < 0 || heads <= 0 || tq < 0 || tkv <= 0) return MIXDQ_ERR_INVALID_ARG;
  if ((out_scale_inv == nullptr) != (out_zero_point == nullptr)) return MIXDQ_ERR_INVALID_ARG;
  if (head_dim != kHeadDim) return MIXDQ_ERR_SHAPE;
  if (batch == 0 || tq == 0) return MIXDQ_OK;       // nothing to write (pointers may be null)
  if (!q || !k || !v || !out) return MIXDQ_ERR_INVALID_ARG;
  const bool quant = out_scale_inv != nullptr;
  // 16-byte vector accesses: rows of 8 halfs, bases and strides multiples of 8 elements
  const int64_t strides[] = {q_batch_stride, q_row_stride, k_batch_stride, k_row_stride,
                             v_batch_stride, v_row_stride, out_batch_stride, out_row_stride};
  for (int64_t s : strides)
    if (s % 8) return MIXDQ_ERR_ALIGNMENT;
  // key-row offsets are formed in 32 bits
  if ((int64_t)tkv * k_row_stride >= (1ll << 31) || (int64_t)tkv * v_row_stride >= (1ll << 31))
    return MIXDQ_ERR_SHAPE;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return MIXDQ_ERR_ALIGNMENT;
  if ((uintptr_t)out & (quant ? 7 : 15)) return MIXDQ_ERR_ALIGNMENT;

  AttnParams p;
  p.q = (const __half*)q; p.k = (const __half*)k; p.v = (const __half*)v; p.out = out;
  p.q_bs = q_batch_stride; p.q_rs = q_row_stride;
  p.k_bs = k_batch_stride; p.k_rs = k_row_stride;
  p.v_bs = v_batch_stride; p.v_rs = v_row_stride;
  p.o_bs = out_batch_stride; p.o_rs = out_row_stride;
  p.tq = tq; p.tkv = tkv; p.heads = heads;
  p.scale_log2 = softmax_scale * 1.4426950408889634f;
  p.s_inv = out_scale_inv; p.zp = out_zero_point;
  p.unfused = (flags & MIXDQ_FLAG_UNFUSED) ? 1 : 0;
  p.attn_blocks = 0; p.pf_blocks = 0; p.n_pf = 0; p.pf_nt = 0; p.pf_delay = 0;
  static const int xcd_on = [] { const char* e = getenv("MIXDQ_ATTN_XCD"); return !(e && e[0] == '0'); }();   // A/B runs
  p.xcd_map = xcd_on;
  for (int i = 0; i < 16; ++i) { p.pf_ptr[i] = nullptr; p.pf_bytes[i] = 0; }
  for (int i = 0; i < n_pf; ++i) {
    if (!pf_ptrs[i] || pf_bytes[i] < 16) continue;
    // the payload is a hint and must never fail a launch: a range that does not start on a 16-byte boundary
    // (a weight view at an odd offset) is rounded inward to the part that 16-byte loads can read
    const uintptr_t b = ((uintptr_t)pf_ptrs[i] + 15) & ~(uintptr_t)15;
    const int64_t cut = (int64_t)(b - (uintptr_t)pf_ptrs[i]);
    if (pf_bytes[i] - cut < 16) continue;
    p.pf_ptr[p.n_pf] = (const char*)b;
    p.pf_bytes[p.n_pf++] = pf_bytes[i] - cut;
  }
  if (p.n_pf) {   // one payload workgroup per CU by default (MIXDQ_PREFETCH_BLOCKS / MIXDQ_PREFETCH_NT: A/B runs)
    static const int blocks = [] { const char* e = getenv("MIXDQ_PREFETCH_BLOCKS"); return e ? atoi(e) : kNumCU; }();
    static const int nt = [] { const char* e = getenv("MIXDQ_PREFETCH_NT"); return e ? atoi(e) : 0; }();
    static const int delay = [] { const char* e = getenv("MIXDQ_PREFETCH_DELAY"); return e ? atoi(e) : 0; }();
    p.pf_blocks = blocks > 0 ? blocks : 0;
    p.pf_nt = nt;
    p.pf_delay = delay >= 0 && delay <= 16 ? delay : 0;
    if (p.pf_blocks == 0) p.n_pf = 0;
  }

  // 128-query workgroups (4 waves x 32 rows) when ONE IMAGE has at least half a chip of them,
  // 64-query ones otherwise.  The rule looks at the image alone, never at the batch: both kernels
  // run the same per-wave arithmetic, and a row of a batch gets exactly the launch geometry it
  // would get alone.  (Round 2's key-split 8-wave kernel for the 1024-token layers at batch 1 is
  // gone: after the loop was pipelined it no longer won -- 16.9 vs 16.2 us, tools/bench_attn.py --
  // and, merging two partial softmaxes, it made a batch-1 result differ in its last bits from the
  // same image inside a batch.)
  const int force = (flags >> 8) & 0xff;         // 4 / 2: waves per workgroup of the pipelined kernel; 1: the short-key kernel
  if (force == 1 && tkv > 2 * kKeys) return MIXDQ_ERR_SHAPE;
  static const bool short_on = [] { const char* e = getenv("MIXDQ_ATTN_SHORT"); return !(e && e[0] == '0'); }();   // A/B runs
  if (force == 1 || (force == 0 && short_on && tkv <= 2 * kKeys)) {   // cross-attention: 77 keys
    p.qblocks = (tq + 127) / 128;
    return launch_attn_short(p, batch, quant, stream);
  }
  const long blocks128 = (long)((tq + 127) / 128) * heads;
  const bool big = force ? force == 4 : blocks128 >= kNumCU / 2;
  p.qblocks = big ? (tq + 127) / 128 : (tq + 63) / 64;
  if (big) return launch_attn<4, 4>(p, batch, quant, stream);
  return launch_attn<2, 4>(p, batch, quant, stream);
}

extern "C" int mixdq_attention_f16(const void* q, const void* k, const void* v, void* out,
                                   int batch, int heads, int head_dim, int tq, int tkv,
                                   int64_t q_batch_stride, int64_t q_row_stride,
                                   int64_t k_batch_stride, int64_t k_row_stride,
                                   int64_t v_batch_stride, int64_t v_row_stride,
                                   int64_t out_batch_stride, int64_t out_row_stride,
                                   float softmax_scale, const float* out_scale_inv,
                                   const float* out_zero_point, int flags, mixdq_stream_t stream) {
  return attention_f16_impl(q, k, v, out, batch, heads, head_dim, tq, tkv, q_batch_stride, q_row_stride,
                            k_batch_stride, k_row_stride, v_batch_stride, v_row_stride, out_batch_stride,
                            out_row_stride, softmax_scale, out_scale_inv, out_zero_point, nullptr, nullptr,
                            0, flags, stream);
}

extern "C" int mixdq_attention_f16_prefetch(const void* q, const void* k, const void* v, void* out,
                                            int batch, int heads, int head_dim, int tq, int tkv,
                                            int64_t q_batch_stride, int64_t q_row_stride,
                                            int64_t k_batch_stride, int64_t k_row_stride,
                                            int64_t v_batch_stride, int64_t v_row_stride,
                                            int64_t out_batch_stride, int64_t out_row_stride,
                                            float softmax_scale, const float* out_scale_inv,
                                            const float* out_zero_point, const void* const* prefetch_ptrs,
                                            const int64_t* prefetch_bytes, int n_prefetch, int flags,
                                            mixdq_stream_t stream) {
  return attention_f16_impl(q, k, v, out, batch, heads, head_dim, tq, tkv, q_batch_stride, q_row_stride,
                            k_batch_stride, k_row_stride, v_batch_stride, v_row_stride, out_batch_stride,
                            out_row_stride, softmax_scale, out_scale_inv, out_zero_point, prefetch_ptrs,
                            prefetch_bytes, n_prefetch, flags, stream);
}

#if MIXDQ_STAMP
// diagnostic builds only: register (or clear, with null) the attention stamp buffer, [grid][4 waves][16] uint64
extern "C" int mixdq_debug_stamps_attn(void* buffer) {
  unsigned long long b = (unsigned long long)(uintptr_t)buffer;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &b, sizeof(b)) == hipSuccess ? MIXDQ_OK : MIXDQ_ERR_LAUNCH;
}
#endif
