/*
 * mixdq_hip.h -- C-ABI of libmixdq_hip.so, the MI355X (gfx950) implementation of the
 * MixDQ W8A8 operator stack (quantize -> INT8 GEMM / implicit-GEMM conv -> dequant epilogue).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types, no libtorch link.
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference checkout, kernels/mixdq_extension/...).  The Python shim that turns these back into
 * the reference's `mixdq_extension._C` functions is mixdq_amd/_C.py; the binding a reference
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions (same as the reference's pybind layer):
 *   - all pointers are DEVICE pointers unless stated otherwise; inputs are borrowed;
 *   - outputs are caller-allocated (the Python shim uses torch.empty so the caching allocator
 *     and graph capture keep working);
 *   - every call is an asynchronous launch on `stream` (a hipStream_t passed as void*);
 *     no host synchronisation, no allocation, scalars (scale_inv, zero_point) are read on the
 *     device  =>  safe under hipGraph capture;
 *   - return value: MIXDQ_OK (0) or an error code; the shim raises RuntimeError with
 *     mixdq_status_string(code) (reference: TORCH_CHECK -> RuntimeError).
 *   - no global mutable state that a result depends on: what the library keeps per process is a zero
 *     page, the GELU table (read-only once built: by the first GEMM+GEGLU call, or mixdq_gelu_table()),
 *     and one "already set up on this device" flag per kernel instantiation (the > 64 KiB LDS opt-in)
 *     and for that table.  Entry points may be called from several threads on different devices.
 */
#ifndef MIXDQ_HIP_H_
#define MIXDQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: status codes 5..9 returned where 1 returned MIXDQ_ERR_UNSUPPORTED; structs and
 * entry points added since (grouped / GEGLU / attention launches, FP16 layers, producer fusions).
 * 3: mixdq_qlinear_w8a8_geglu takes its weight rows in value|gate groups of 16 (was 32) and N % 32.
 * Bumped whenever an existing entry point changes its signature, operand layout or error behaviour. */
#define MIXDQ_ABI_VERSION 3

typedef void* mixdq_stream_t; /* hipStream_t */

enum mixdq_status {
  MIXDQ_OK = 0,
  MIXDQ_ERR_INVALID_ARG = 1,     /* null pointer, negative size, ndim out of range ...            */
  MIXDQ_ERR_ALIGNMENT = 2,       /* "Int8 kernel with input or output alignment not to 4 is not
                                    supported." (qlinear.cc:131-134, qconv2d.cc:200-203)          */
  MIXDQ_ERR_UNSUPPORTED = 3,     /* dilation != 1 (op/qconv2d.py:120 "dilation has bugs")         */
  MIXDQ_ERR_LAUNCH = 4,          /* hipGetLastError() != hipSuccess ("CUTLASS kernel failed")    */
  /* codes 5.. have no reference counterpart: limits of entry points the reference does not have */
  MIXDQ_ERR_W4_SHAPE = 5,        /* MIXDQ_FLAG_W4: K % 32 != 0 (conv: C % 32) or an operand pointer
                                    not 16-byte aligned                                           */
  MIXDQ_ERR_GEGLU_SHAPE = 6,     /* mixdq_qlinear_w8a8_geglu: N % 32, K % 16 or pointer alignment */
  MIXDQ_ERR_PADDING = 7,         /* conv: padding >= kernel size (a window with no in-image tap)  */
  MIXDQ_ERR_ROWMAP_RESIDUAL = 8, /* output row map and residual in one call                       */
  MIXDQ_ERR_SHAPE = 9            /* shape outside a fused kernel's range (head_dim != 64, GroupNorm
                                    geometry, LayerNorm width, tensor rank)                       */
};

/* Bit flags accepted by the compute entry points. */
enum mixdq_flags {
  /* Epilogue / quantize rounding variant.  Default (0) = variant A: the
     multiply-add is one fused FMA (what nvcc -fmad=true makes of the reference's mul+add).
     MIXDQ_FLAG_UNFUSED = variant B: round after the multiply, then add.                          */
  MIXDQ_FLAG_UNFUSED = 1,
  /* The weight operand of mixdq_qlinear_w8a8_rows / mixdq_qconv2d_w8a8[_table] is PACKED signed
     4-bit ("nibble-planar per 8": byte j of each 4-byte group holds k[8g+j] in the high nibble and
     k[8g+4+j] in the low nibble, two's complement; [N, K/2] bytes, conv [K,R,S,C/2]).  Needs
     K % 32 == 0 (conv: C % 32 == 0).  No reference counterpart: the reference's 4-bit layers
     fall back to FP16 (nn/Linear.py:31,133-134); results equal the W8 path run on the unpacked
     values.  bias0 / scale are those of the unpacked integers. */
  MIXDQ_FLAG_W4 = 2,
  /* mixdq_qconv2d_w8a8_table only: X is [N, H/2, W/2, C] and the conv runs on its nearest-neighbour
     2x upsampling (H, W stay the conv's input size) -- Upsample2D's conv(interpolate(x, 2.0)) without
     the upsampled tensor (quantizing commutes with nearest upsampling: the INT8 values are the same).
     3x3 / stride 1 / pad 1 shapes of the LDS-halo kernel only (mixdq_conv_halo_select != 0), else
     MIXDQ_ERR_SHAPE.  No reference counterpart. */
  MIXDQ_FLAG_UPSAMPLE2X = 4,
  /* mixdq_qlinear_f16in_w8a8 only: the FP16 operand's rows follow the output row map (see there). */
  MIXDQ_FLAG_A_ROWMAP = 8
  /* bits 8..15: force a kernel configuration id (tuning / tests); 0 = automatic */
};

const char* mixdq_status_string(int status);
int mixdq_abi_version(void);
/* sha256 (first 16 hex digits) over the kernel sources this library was built from (mixdq_amd/build.py writes it
 * at build time): lets a run say which sources the LOADED library carries -- the provenance check of bench.py's
 * counter columns -- whatever MIXDQ_HIP_LIB points at. */
const char* mixdq_build_csrc_sha16(void);

/* ---------------------------------------------------------------------------------------------
 * a1. FP16 -> INT8 per-tensor affine quantize.
 * Replaces: quantize_per_tensor_to_int8 / quantize_per_tensor_to_int8_vectorized
 *           (csrc/quant_dequant/quantize.cc:9-53, quantize_kernel.cu:10-48,
 *            quantize_kernel_vectorized.cu:29-95).  One kernel serves both names.
 *   q = (int8) clamp(rint(fma(f32(x), *scale_inv, *zero_point)), -128, 127)
 * `sizes`/`x_strides`/`out_strides` are HOST arrays of length ndim (1..8), strides in elements.
 * The reference reads x linearly and ignores strides (quantize_kernel.cu:20-25), which is only
 * right for dense inputs; this entry point implements the intended strided semantics.
 */
int mixdq_quantize_f16_i8(const void* x_f16, int8_t* out,
                          const int64_t* sizes, const int64_t* x_strides,
                          const int64_t* out_strides, int ndim,
                          const float* scale_inv, const float* zero_point,
                          int flags, mixdq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a2. INT8 x INT8 -> INT32 GEMM with FP32 epilogue -> FP16.
 * Replaces: qlinear_w8_a8_ohalf (csrc/qlinear/qlinear.cc:13-137) and the four CUTLASS kernels
 *           cutlassGemm_{withBias,noBias}_{optimal,small}Alignment.cu.
 *   D[m,n] = f16( fma( (f32(sum_k A[m,k]*W[n,k]) - bias0[n]), scale[n], f32(bias[n]) ) )
 *            (no bias: f16((acc - bias0[n]) * scale[n]))
 * A: [M,K] row-major (lda = K), W: [N,K] row-major, D: [M,N] row-major (ldd = N).
 * Alignment: K % 4 == 0 and N % 4 == 0, else MIXDQ_ERR_ALIGNMENT (qlinear.cc:96-134).
 * The reference's unused arguments (weight_scale, input_scale, input_zero_point,
 * weight_sum_by_input_channels) stay in the Python signature and are not passed down.
 */
int mixdq_qlinear_w8a8(const int8_t* A, const int8_t* W,
                       const float* bias0, const float* scale,
                       const void* bias_f16_or_null, void* D_f16,
                       int64_t M, int N, int K,
                       int flags, mixdq_stream_t stream);

/* Same GEMM with two extensions (no reference counterpart; both keep results bit-identical to the
 * unfused sequence of reference ops + torch ops):
 *  - an output row map, used by QuantizedLinear's BOS path (nn/Linear.py:178-194) to write rows
 *    straight into the [B, T, N] result instead of torch.cat:
 *      D_row(m) = (m / group_rows) * group_stride + group_offset + (m % group_rows)
 *    (group_rows = T-1 = 76, group_stride = T = 77, group_offset = 1).  group_rows <= 0 = identity;
 *  - a residual added after the epilogue's FP16 rounding, exactly as a following torch half add:
 *      D[m,n] = f16( f32(f16(epilogue)) + f32(residual[(m / residual_row_div) * N + n]) )
 *    residual_row_div = 1: a full [M,N] residual (x + attn(x), x + ff(x)); = P*Q: one row per
 *    image (h + time_emb[:, :, None, None] after a conv).  Not combinable with a row map.
 */
int mixdq_qlinear_w8a8_rows(const int8_t* A, const int8_t* W,
                            const float* bias0, const float* scale,
                            const void* bias_f16_or_null, void* D_f16,
                            int64_t M, int N, int K,
                            int group_rows, int group_stride, int group_offset,
                            const void* residual_f16_or_null, int64_t residual_row_div,
                            int flags, mixdq_stream_t stream);

/* a1 + a2 in ONE launch: the GEMM quantizes its own activation operand ("A8 quant fused into INT8 GEMM").
 * Replaces the reference's pair of launches per layer -- quant_op(x, act_scales_inv, act_zero_points) followed
 * by qlinear (nn/Linear.py:162-176); 1x1 / pad-0 convs on NHWC rows too (nn/Conv2d.py:294-311) -- with results
 * bit-identical to mixdq_quantize_f16_i8 -> mixdq_qlinear_w8a8_rows:
 *   A_q[m,k] = (int8) clamp(rint(fma(f32(A[m * lda + k]), *act_scale_inv, *act_zero_point)), -128, 127)
 *   D        = the a2 epilogue over A_q (row map, residual and MIXDQ_FLAG_W4 / _UNFUSED as above).
 * A_f16: FP16, row m at element offset m * lda (lda >= K, lda % 8 == 0, 16-byte aligned base: a dense
 * [M, K] matrix, or a column slice x[:, c0:c0+K] of a wider row-major / NHWC tensor).  MIXDQ_FLAG_A_ROWMAP:
 * the operand's row m is row (m / group_rows) * group_stride + group_offset + m % group_rows of A_f16 --
 * the same map as the output's, i.e. the BOS slice x[:, 1:, :] of a [B, T, K] tensor read in place.
 * Range: K a multiple of the chosen tile's K depth (128; 64 on the 128x128 / 256x128 tiles), operands
 * 16-byte aligned, the FP16 operand below 4 GiB; MIXDQ_ERR_SHAPE otherwise (the caller then issues the
 * two launches).  mixdq_qlinear_f16in_supported() answers that question without launching. */
int mixdq_qlinear_f16in_w8a8(const void* A_f16, int64_t lda,
                             const float* act_scale_inv, const float* act_zero_point,
                             const int8_t* W, const float* bias0, const float* scale,
                             const void* bias_f16_or_null, void* D_f16,
                             int64_t M, int N, int K,
                             int group_rows, int group_stride, int group_offset,
                             const void* residual_f16_or_null, int64_t residual_row_div,
                             int flags, mixdq_stream_t stream);
/* 1 if mixdq_qlinear_f16in_w8a8 takes this problem (sizes only; pointers must still be 16-byte aligned),
 * else 0.  `rows_addressed`: rows of A_f16 the operand spans (M, or the mapped extent under a row map). */
int mixdq_qlinear_f16in_supported(int64_t M, int N, int K, int64_t lda, int64_t rows_addressed, int w4);
/* 1 if the one launch is expected to be cheaper than mixdq_quantize_f16_i8 + mixdq_qlinear_w8a8_rows for this
 * problem (a cost model fitted to MI355X measurements, csrc/igemm_aq.hip): every workgroup quantizes the rows of
 * its own tile, so wide layers at small M repeat the quantizer's arithmetic N / BN times and lose to the
 * stand-alone kernel.  mixdq_amd's modules fuse where this says 1 (MIXDQ_F16IN=1: wherever supported; 0: never). */
int mixdq_qlinear_f16in_preferred(int64_t M, int N, int K, int w4);
/* Configuration id (MIXDQ_IGEMM_CONFIGS) the automatic choice makes for such a launch; -1 = unsupported. */
int mixdq_qlinear_f16in_select_id(int64_t M, int N, int K, int w4);

/* ---------------------------------------------------------------------------------------------
 * a3 + a4. INT8 NHWC implicit-GEMM conv2d (cross-correlation) with the same epilogue, and the
 * activation zero-point propagation for padded convs.
 * Replaces: qconv2d_w8_a8_ohalf (csrc/qconv2d/qconv2d.cc:27-206), the eight CUTLASS conv
 *           kernels cutlassConv2d_*.cu, and activation_zero_point_propagate
 *           (csrc/qconv2d/conv_act_zero_point_propagate.cu:11-83).
 * X: [N,H,W,C] int8 (channels-last), Wt: [K,R,S,C] int8 (channels-last weight),
 * D: [N,P,Q,K] f16 (channels-last), P = (H + 2*pad - R)/stride + 1 (same for Q).
 *   pad == 0: bias0[K] required (per-channel);  wsum ignored.
 *   pad  > 0: wsum[K,R,S] f32 and *zero_point required; per output pixel
 *             bias0[n,p,q,k] = f32(sum over in-bounds taps of wsum[k,r,s]) * zp.
 *             The reference materialises that as an [N,P,Q,K] f32 tensor on every call; here a
 *             small table of tap-subset sums (mixdq_conv_border_table) is built into `workspace`
 *             on the same stream and the conv epilogue looks its border class up per pixel.
 * `workspace`: device buffer of at least mixdq_qconv2d_workspace_bytes(K,R,S,pad) bytes
 *             (may be null when that is 0).
 * dilation must be 1 (MIXDQ_ERR_UNSUPPORTED otherwise); C % 4 == 0 and K % 4 == 0.
 */
size_t mixdq_qconv2d_workspace_bytes(int K, int R, int S, int pad);

int mixdq_qconv2d_w8a8(const int8_t* X_nhwc, const int8_t* Wt_krsc,
                       const float* scale,
                       const float* wsum_krs_or_null, const float* zero_point,
                       const float* bias0_or_null,
                       const void* bias_f16_or_null, void* D_nhwc_f16,
                       void* workspace,
                       int N, int H, int W, int C, int K, int R, int S,
                       int stride, int pad, int dilation,
                       int flags, mixdq_stream_t stream);

/* The two halves of the call above, for callers that cache the table per layer
 * (mixdq_amd.nn.QuantizedConv2d does: the table depends only on the weights). */
int mixdq_conv_border_table(const float* wsum_krs, float* table,
                            int K, int R, int S, mixdq_stream_t stream);

int mixdq_qconv2d_w8a8_table(const int8_t* X_nhwc, const int8_t* Wt_krsc,
                             const float* scale,
                             const float* table_or_null, const float* zero_point,
                             const float* bias0_or_null,
                             const void* bias_f16_or_null, void* D_nhwc_f16,
                             int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad,
                             const void* residual_f16_or_null, int64_t residual_row_div,
                             int flags, mixdq_stream_t stream);

/* Stand-alone restatement of the reference's materialised zero-point propagation
 * (conv_act_zero_point_propagate.cu:11-83): out[n,p,q,k] f32, NHWC.  Not on the fast path;
 * kept so the a4 row of the scope table has a directly comparable artefact. */
int mixdq_conv_zero_point_propagate(const float* wsum_krs, const float* zero_point,
                                    float* out_npqk,
                                    int N, int H, int W, int K, int R, int S,
                                    int stride, int pad, mixdq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * FP16 debug GEMM.  Replaces: qlinear_fp_reference (csrc/qlinear/qlinear.cc:140-204,
 * cutlassGemm_reference.cu:117-283).  D[M,N] = A[M,K] * B[K,N], B row-major [K,N]
 * (qlinear.cc:161), FP32 accumulate, FP16 out.  Not on the hot path.
 */
int mixdq_gemm_f16(const void* A_f16, const void* B_f16_kn, void* D_f16,
                   int64_t M, int N, int K, mixdq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Producer fusions (no reference counterpart: the reference runs stock
 * PyTorch FP16 ops followed by its quantize kernel).  Each keeps the rounding points of the unfused
 * sequence -- normalised value -> FP16, SiLU / GELU -> FP16, product -> FP16 -- and then applies the
 * a1 quantizer to that FP16 value.  Reduction orders are fixed and the transcendental steps are
 * include/mixdq_math.h, so the oracle reproduces them bit-for-bit.
 *
 * GroupNorm (+SiLU) + quantize on an NHWC tensor x [N, HW, C] (fp16), G groups, gamma/beta fp16 [C].
 * Writes int8 [N, HW, C] (if out_q != null) and/or the fp16 activation (if out_f16 != null).
 * Needs C % 8 == 0 and groups that an 8-channel run straddles at most once
 * (MIXDQ_ERR_SHAPE otherwise).  `workspace`: mixdq_groupnorm_workspace_bytes() bytes. */
size_t mixdq_groupnorm_workspace_bytes(int N, int64_t HW, int C, int G);
int mixdq_groupnorm_silu_quantize(const void* x_nhwc_f16, const void* gamma_f16,
                                  const void* beta_f16, float eps, int apply_silu,
                                  const float* scale_inv, const float* zero_point,
                                  int8_t* out_q_or_null, void* out_f16_or_null, void* workspace,
                                  int N, int64_t HW, int C, int G, int flags,
                                  mixdq_stream_t stream);

/* The same on a channel concatenation that is never made in memory (an up-block's
 * cat([hidden, skip], dim=1)): channels 0 .. C1-1 are read from x [N, HW, C1], the remaining C - C1
 * from x2 [N, HW, C - C1] (C1 % 8 == 0; C1 == C and x2 == null: one source).  Every output is what
 * mixdq_groupnorm_silu_quantize gives on the concatenated tensor. */
int mixdq_groupnorm_silu_quantize2(const void* x_nhwc_f16, int C1, const void* x2_nhwc_f16_or_null,
                                   const void* gamma_f16, const void* beta_f16, float eps,
                                   int apply_silu, const float* scale_inv, const float* zero_point,
                                   int8_t* out_q_or_null, void* out_f16_or_null, void* workspace,
                                   int N, int64_t HW, int C, int G, int flags,
                                   mixdq_stream_t stream);

/* ... and, from the same pass, the INPUT itself quantized (no normalisation): raw_q[i] (HOST arrays
 * of two device pointers; entry 0: source x, entry 1: source x2; null entries / null arrays: none)
 * receives mixdq_quantize_f16_i8(source i; raw_scale_inv[i], raw_zero_point[i]) in that source's own
 * [N, HW, Cs] layout -- the operand of a layer that reads the same activation as the norm (the ResNet
 * block's 1x1 conv_shortcut beside norm1; a split shortcut quantizes each half with its own
 * quantizer, nn/Conv2d.py:330-343). */
int mixdq_groupnorm_silu_quantize3(const void* x_nhwc_f16, int C1, const void* x2_nhwc_f16_or_null,
                                   const void* gamma_f16, const void* beta_f16, float eps,
                                   int apply_silu, const float* scale_inv, const float* zero_point,
                                   int8_t* out_q_or_null, void* out_f16_or_null,
                                   const float* const* raw_scale_inv,
                                   const float* const* raw_zero_point, int8_t* const* raw_q,
                                   void* workspace, int N, int64_t HW, int C, int G, int flags,
                                   mixdq_stream_t stream);

/* LayerNorm over the last dimension of x [M, C] (fp16) + up to three quantizers of the same
 * normalised FP16 value (to_q / to_k / to_v have their own activation scales).  HOST arrays of
 * n_out device pointers.  C % 16 == 0, C <= 2048 (MIXDQ_ERR_SHAPE otherwise).  Row statistics in the
 * tiling-independent order of oracle/mixdq_oracle.c (per-16-column partials, Chan's combination), so that
 * mixdq_qlinear_w8a8_ln -- the same LayerNorm in the epilogue of the GEMM that produces x -- gives the same bits. */
int mixdq_layernorm_quantize(const void* x_f16, const void* gamma_f16, const void* beta_f16,
                             float eps, int64_t M, int C, int n_out,
                             const float* const* scale_inv, const float* const* zero_point,
                             int8_t* const* out_q, void* out_f16_or_null, int flags,
                             mixdq_stream_t stream);

/* a2 + residual + LayerNorm + quantize in ONE launch: the GEMM of mixdq_qlinear_w8a8_rows (bias, residual as
 * there; no row map) whose output rows D [M, N] are ALSO normalised over their N columns and quantized for up to
 * three consumers, exactly as mixdq_layernorm_quantize(D, ...) would do in a second launch:
 *   out_q[i][m, n] = quantize_i( f16( fma((D[m,n] - mean_m) * rstd_m, gamma[n], beta[n]) ) ),  out_f16 = that FP16.
 * A column tile of 80 columns is exactly one unit of the LayerNorm's reduction order (oracle/mixdq_oracle.c):
 * every tile publishes ONE 16-byte record per row -- {sum, centred sum of squares, launch tag} -- into
 * `workspace` with a write-through store, polls the N / 80 records of each of its rows until they carry this
 * launch's tag, combines them and normalises the columns it still holds in LDS; the bits equal the two-launch
 * chain.  Range: N = 1280 or 640 (N / 80 a power of two <= 16 that equals the LayerNorm's unit count),
 * K % 128 == 0, and the whole launch resident at once -- at most one 64 x 80 (or 128 x 80) tile per CU:
 * M <= 1024 at N = 1280 (2048 on 128-row tiles), M <= 4096 at N = 640 -- MIXDQ_ERR_SHAPE otherwise
 * (mixdq_qlinear_ln_select_id() < 0; the caller then issues the two launches).
 * `workspace`: mixdq_qlinear_ln_workspace_bytes(M, N) bytes, 16-byte aligned, ZERO before the first use (its
 * first page holds a launch counter that tags the records: a launch of any shape may follow on the same
 * buffer).  One launch at a time per workspace (launches of one stream are) AND per device: the tiles of a row
 * block wait for each other, so the whole grid must be resident at once -- the entry point checks the grid against
 * the device's real CU count and the runtime's occupancy answer (MIXDQ_ERR_SHAPE if it does not fit), but two such
 * launches on DIFFERENT streams of one device can each hold CUs the other needs: do not run them concurrently.  A
 * workgroup that gives up waiting (bounded spin) writes its rows as NaN and sets a sticky error word in the
 * workspace; mixdq_qlinear_ln_status() reads it (0 = every launch on this workspace found all its records).
 * No reference counterpart (stock nn.LayerNorm + quant_op, nn/Linear.py:162-164). */
size_t mixdq_qlinear_ln_workspace_bytes(int64_t M, int N);
int mixdq_qlinear_ln_select_id(int64_t M, int N, int K);   /* tile id (44, 45, 56) or -1 = not supported */
int mixdq_qlinear_ln_status(const void* workspace, int* status, mixdq_stream_t stream);   /* blocking; not in a capture */
int mixdq_qlinear_w8a8_ln(const int8_t* A, const int8_t* W, const float* bias0, const float* scale,
                          const void* bias_f16_or_null, void* D_f16, int64_t M, int N, int K,
                          const void* residual_f16_or_null, int64_t residual_row_div,
                          const void* gamma_f16, const void* beta_f16, float eps, int n_out,
                          const float* const* scale_inv, const float* const* zero_point,
                          int8_t* const* out_q, void* out_f16_or_null, void* workspace,
                          int flags, mixdq_stream_t stream);

/* GEGLU + quantize: h [M, 2D] fp16 (ff.net.0.proj output) -> fp16(h[:, :D] * fp16(gelu(h[:, D:])))
 * -> int8 [M, D] and/or fp16 [M, D].  D % 8 == 0. */
int mixdq_geglu_quantize(const void* h_f16, int64_t M, int D,
                         const float* scale_inv, const float* zero_point,
                         int8_t* out_q_or_null, void* out_f16_or_null, int flags,
                         mixdq_stream_t stream);

/* to_q + cross-attention in ONE launch: the INT8 GEMM of attn2.to_q (no bias) whose fp16 result --
 * 64 query rows x two heads per workgroup -- never leaves the chip: it is multiplied against the
 * (at most 128) keys / values of those heads with the arithmetic of mixdq_attention_f16, and the
 * attention output is written as to_out.0's INT8 operand (out_scale_inv given) or as fp16.
 * Bit-identical to mixdq_qlinear_w8a8 followed by mixdq_attention_f16.  No reference counterpart.
 * A [M, K] int8 (M = images x rows_per_image, rows_per_image % 64 == 0), W [N, K] (N % 128 == 0,
 * heads = N / 64, K % 128 == 0), k / v fp16 [images, tkv, N] with the given strides in elements
 * (multiples of 8; column slices of a packed k|v projection are fine), tkv <= 128. */
int mixdq_qlinear_w8a8_attn(const int8_t* A, const int8_t* W, const float* bias0, const float* scale,
                            const void* k_f16, const void* v_f16, void* out, int64_t M, int N, int K,
                            int rows_per_image, int tkv, int64_t k_batch_stride, int k_row_stride,
                            int64_t v_batch_stride, int v_row_stride, float softmax_scale,
                            const float* out_scale_inv_or_null, const float* out_zero_point_or_null,
                            int flags, mixdq_stream_t stream);

/* Grouped form of mixdq_qlinear_w8a8_rows: `ngroups` independent Linears that read the SAME int8
 * activations A [M, K] -- the 70 cross-attention k|v projections of the text embeddings, the 22
 * time-embedding projections -- in ONE launch (gridDim.y = member).  No reference counterpart (it
 * launches one GEMM per layer); every member's outputs are bit-identical to its own launch.
 * `groups_device`: DEVICE array of members; N may differ per member (max_N = the largest).  The
 * row map is shared.  K % 16 == 0 (W4: % 32) and N % 4 == 0. */
typedef struct mixdq_gemm_group {
  const int8_t* W;               /* [N, K] (MIXDQ_FLAG_W4: [N, K/2] packed) */
  const float* bias0;            /* [N] */
  const float* scale;            /* [N] */
  const void* bias_f16_or_null;  /* [N] */
  void* D_f16;                   /* output base */
  int32_t N;
  int32_t reserved;
} mixdq_gemm_group;
int mixdq_qlinear_w8a8_grouped(const int8_t* A, const mixdq_gemm_group* groups_device, int ngroups,
                               int64_t M, int max_N, int K, int group_rows, int group_stride,
                               int group_offset, int flags, mixdq_stream_t stream);

/* ff.net.0.proj + GEGLU + quantize in one launch: the GEMM of mixdq_qlinear_w8a8 whose N = 2D
 * output columns arrive as value|gate groups of 16 ([v 0..15 | g 0..15 | v 16..31 | g 16..31 ...]:
 * the caller stores W, bias0, scale and bias with rows in that order -- then every 32-column MFMA
 * tile holds the value and the gate of 16 outputs, in the same lanes), reduced in the epilogue to
 * out[m, d] = sat8(rint(y * scale_inv + zero_point)), y = f16(f16(v) * f16(gelu(f16(g)))) -- every
 * rounding point of GEMM -> fp16 -> mixdq_geglu_quantize is kept, so the int8 [M, D] result is
 * bit-identical to that two-launch chain (diffusers GEGLU: hidden, gate = proj(x).chunk(2);
 * hidden * gelu(gate)).  N % 32 == 0, K % 16 == 0 (MIXDQ_ERR_GEGLU_SHAPE otherwise). */
int mixdq_qlinear_w8a8_geglu(const int8_t* A, const int8_t* W_interleaved, const float* bias0,
                             const float* scale, const void* bias_f16_or_null, int8_t* out_i8,
                             int64_t M, int N, int K, const float* out_scale_inv,
                             const float* out_zero_point, int flags, mixdq_stream_t stream);

/* FP16 attention core, head_dim 64: out[b, t, h*64 + d] = softmax_k(q . k * softmax_scale) v, per
 * head h.  The reference keeps these matmuls in FP16 (quant_block.py:630-637: get_attention_scores
 * + torch.bmm; diffusers' AttnProcessor at run time) — only to_q/to_k/to_v/to_out.0 are quantized
 * Linears — so this is a floating-point op with a tolerance oracle, not an integer one.
 * q [batch, tq, heads*64], k/v [batch, tkv, heads*64] fp16 with arbitrary batch/row strides in
 * ELEMENTS (multiples of 8; column slices of a fused q|k|v projection are fine).  Softmax in FP32,
 * P rounded to FP16 for the second product, output rounded to FP16.
 * out: fp16 rows (out_scale_inv == null), or — fused producer of to_out.0's INT8 operand — int8
 * rows q = sat8(rint(f16_out * scale_inv + zero_point)) (same arithmetic as mixdq_quantize_f16_i8).
 * flags: MIXDQ_FLAG_UNFUSED selects the unfused quantize variant; bits 8..15 force the workgroup
 * shape (4 = 128 query rows, 2 = 64). */
int mixdq_attention_f16(const void* q_f16, const void* k_f16, const void* v_f16, void* out,
                        int batch, int heads, int head_dim, int tq, int tkv,
                        int64_t q_batch_stride, int64_t q_row_stride,
                        int64_t k_batch_stride, int64_t k_row_stride,
                        int64_t v_batch_stride, int64_t v_row_stride,
                        int64_t out_batch_stride, int64_t out_row_stride,
                        float softmax_scale, const float* out_scale_inv_or_null,
                        const float* out_zero_point_or_null, int flags, mixdq_stream_t stream);

/* mixdq_attention_f16 with a PREFETCH payload: beside the attention workgroups the launch carries
 * workgroups that do nothing but read the `n_prefetch` (<= 16) given byte ranges (HOST arrays of device
 * pointers / sizes) -- the INT8 weights of the layers that FOLLOW this attention in the network.  At
 * batch 1 a 1024-token self-attention occupies 62 % of the SIMDs for ~17 us and leaves HBM idle, while
 * every GEMM behind it streams weights that were last touched a step ago (2.6 GB per step: nothing
 * survives in the 256 MB Infinity Cache); read here, they are served from that cache when their GEMM
 * runs.  The payload has no effect on any result.  No reference counterpart.
 * Ranges may be null / empty; the attention itself is mixdq_attention_f16's, bit for bit.  Launches on
 * the short-key kernel (tkv <= 128) ignore the payload. */
int mixdq_attention_f16_prefetch(const void* q_f16, const void* k_f16, const void* v_f16, void* out,
                                 int batch, int heads, int head_dim, int tq, int tkv,
                                 int64_t q_batch_stride, int64_t q_row_stride,
                                 int64_t k_batch_stride, int64_t k_row_stride,
                                 int64_t v_batch_stride, int64_t v_row_stride,
                                 int64_t out_batch_stride, int64_t out_row_stride,
                                 float softmax_scale, const float* out_scale_inv_or_null,
                                 const float* out_zero_point_or_null,
                                 const void* const* prefetch_ptrs, const int64_t* prefetch_bytes,
                                 int n_prefetch, int flags, mixdq_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * FP16 layers.  Replaces the reference's FP fallback for layers without an activation quantizer
 * or with unsupported weight bits -- F.linear / F.conv2d on the FP16 weight (nn/Linear.py:155-156,
 * nn/Conv2d.py:306-309), i.e. cuBLAS / cuDNN there -- with the same LDS-DMA / MFMA kernel family
 * on FP16 operands (v_mfma_f32_32x32x16_f16, FP32 accumulation), so the captured step contains no
 * vendor-library kernel.  Floating point: a tolerance oracle (FP32 reference), not bit parity.
 *   D[m,n] = f16( sum_k f32(A[m,k]) * f32(W[n,k]) + f32(bias[n]) )  [+ residual, as above]
 * A [M,K], W [N,K], D [M,N] row-major fp16; conv: X [N,H,W,C], Wt [K,R,S,C], D [N,P,Q,K].
 * K % 8 == 0 (conv: C % 8 == 0) and N % 4 == 0 run on MFMA tiles, anything else (conv_in: C = 4)
 * on a one-output-per-thread kernel.  flags: bits 8..15 force a tile configuration. */
int mixdq_linear_f16(const void* A_f16, const void* W_f16, const void* bias_f16_or_null,
                     void* D_f16, int64_t M, int N, int K, const void* residual_f16_or_null,
                     int64_t residual_row_div, int flags, mixdq_stream_t stream);
int mixdq_conv2d_f16(const void* X_f16, const void* Wt_f16, const void* bias_f16_or_null,
                     void* D_f16, int N, int H, int W, int C, int K, int R, int S, int stride,
                     int pad, const void* residual_f16_or_null, int64_t residual_row_div,
                     int flags, mixdq_stream_t stream);

/* Which kernel instantiation mixdq_qlinear_w8a8 / mixdq_qconv2d_w8a8 will launch for a problem of
 * M rows x N output channels (k_align = K for Linear, C for Conv2d; k_total = K or R*S*C): the block tile BM x BN x BK
 * and LDS stage count of `igemm_kernel<BM,BN,BK,STAGES,CONV>`, or zeros for the small-alignment
 * generic kernel.  Host-only
 * query, used by bench.py to attribute measured time to kernel names.  No reference counterpart. */
int mixdq_igemm_select(int64_t M, int N, int k_align, int k_total, int* bm, int* bn, int* bk,
                       int* stages);
/* The configuration id behind mixdq_igemm_select (the ids of MIXDQ_IGEMM_CONFIGS in csrc/igemm.hip,
 * which bits 8..15 of `flags` can force; 0 = the small-alignment generic kernel, -1 = invalid):
 * two ids may share a tile shape and differ in the number of waves working on it. */
int mixdq_igemm_select_id(int64_t M, int N, int k_align, int k_total);
/* The same for a packed-W4 weight operand (MIXDQ_FLAG_W4); -1 = invalid (k_align % 32 != 0). */
int mixdq_igemm_select_id_w4(int64_t M, int N, int k_align, int k_total);
/* The same for the GEMM + GEGLU + quantize launch (mixdq_qlinear_w8a8_geglu), whose tiles hold whole
 * 32-column value | gate groups (BN % 32 == 0); from 1.5 workgroups of 256x256 per CU on it runs on the
 * four-phase 256x256 tile (id 70) like a plain Linear. */
int mixdq_igemm_select_id_geglu(int64_t M, int N, int k_total, int w4);

/* Tile id of the LDS-resident-halo kernel (csrc/iconv.hip) that mixdq_qconv2d_w8a8[_table] runs this
 * INT8 conv on when no tile is forced -- 90: 8 x 16 output pixels x 80 channels per workgroup, 91:
 * 8 x 8 x 80, 92: 16 x 16 x 80 (64-byte channel chunks), 93: 16 x 16 x 160 (K % 160 == 0; the choice from
 * batch 8 on) -- or 0 when the implicit-GEMM family runs it (not 3x3 / stride 1 / pad 1, C % 64 != 0,
 * H or W % 8 != 0, packed-W4 weights).  Ids 90 .. 93 can be forced through bits 8..15 of `flags`. */
int mixdq_conv_halo_select(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);

/* Introspection (tests): the table the GEMM + GEGLU epilogue of the large tiles looks GELU up in --
 * 2 x 0x4c00 uint16: f16 bits of gelu(g) for the FP16 gate g with bits (sign << 15) | magnitude,
 * magnitude < 0x4c00 (|g| < 16), positive half first -- copied to `out_device` (77 824 bytes). */
int mixdq_gelu_table(uint16_t* out_device, mixdq_stream_t stream);
/* The FP16 -> FP16 SiLU table the GroupNorm apply pass of LARGE launches looks SiLU up in (round 6; built on first use
 * by the specification itself, include/mixdq_math.h: same bits as the arithmetic): n_pos entries for y = +bits, then
 * n_neg for y = -bits; values outside take the arithmetic.  Builds the table if needed; copies it to out_device
 * (2 * (n_pos + n_neg) bytes) unless null.  No reference counterpart (stock nn.SiLU). */
int mixdq_silu_table(uint16_t* out_device_or_null, int* n_pos, int* n_neg, mixdq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIXDQ_HIP_H_ */
