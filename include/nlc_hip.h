/*
 * nlc_hip.h — C ABI of libnlc_hip.so, the MI355X (gfx950) kernel library behind the
 * DDIM/EDM + noise-level-correction (NLC) sampling hot path.
 *
 * The reference (Walleclipse/Diffusion-NLC) is pure Python on PyTorch and has no FFI
 * of its own; every entry point below replaces a stock torch.nn.functional call (or a
 * short chain of them) that the reference makes on the sampling path.  The reference
 * call site each one stands in for is cited as  file:line  under /root/reference.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative NLC_E* code; nlc_last_error()
 *    gives a thread-local message.  Nothing allocates, nothing synchronises the host.
 *  - All pointers are BORROWED device pointers (hipMalloc'ed / torch caching allocator).
 *    The caller keeps them alive until `stream` has passed the launch.
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *  - dtype = storage type of activations and packed weights: NLC_F32 (exact f32 MFMA path), NLC_BF16 or NLC_F16
 *    (16-bit operands, f32 accumulate; v_mfma_f32_16x16x32_{bf16,f16} run at the same rate on gfx950).  NLC_F16 is the
 *    reference's own half-precision mode (src/fp16_util.py:15-22, src/unet_adm.py:620-634).  On NLC_F32 tensors the
 *    convolutions can additionally run in NLC_MATH_F16X3 (nlc_conv_desc.math): f32 storage, split-f16 matrix math.
 *  - Activations are channels-last: [B][H][W][C] ("NHWC"); token tensors are [B][T][C].
 *    Weights are pre-packed by the host into [Cout_pad][KH*KW][Cin_pad] (see
 *    nlc_conv_pack_dims) in the compute dtype.
 */
#ifndef NLC_HIP_H
#define NLC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLC_ABI_VERSION 7

enum { NLC_F32 = 0, NLC_BF16 = 1, NLC_F16 = 2 };
/* matrix arithmetic of nlc_conv2d on NLC_F32 tensors (nlc_conv_desc.math; weights must be packed for the same mode):
 *   NLC_MATH_NATIVE  the dtype's own MFMA (f32: exact v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit rate)
 *   NLC_MATH_F16X3   every f32 operand x is split into two halves, hi = f16(x), lo = f16(x - hi), and a product is
 *                    hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with f32 accumulate: 3/16 of the 16-bit rate instead of
 *                    1/16.  Storage, bias, embedding, residual, activation and every non-convolution kernel stay exact f32.
 *                    Accuracy and domain (f16 has 11 significand bits, normal range 2^-14 .. 65504, subnormal spacing 2^-24):
 *                      weights      are scaled per OUTPUT ROW by a power of two at pack time so that the row's largest |w| lies in
 *                                   [2^14, 2^15) (nlc_pack_conv_weights_ex writes the inverse factors to w_scale_out, the
 *                                   convolution multiplies its sum by nlc_conv_desc.w_scale[n] - exact): hi + lo carries 22
 *                                   significand bits of every weight within 2^-17 of its row's maximum, and an absolute error
 *                                   <= 2^-39 of that maximum below; no weight can overflow or lose its hi half.
 *                      activations  are split unscaled, in the kernel: 22 significand bits for 2^-3 <= |x| < 65504, absolute
 *                                   error <= 2^-25 below 2^-3 (lo is an f16 subnormal there) - the inputs of this path are
 *                                   GroupNorm / SiLU outputs and residual streams of magnitude O(1..100).  |x| >= 65504 is
 *                                   OUTSIDE the domain: the kernels run with the FP16_OVFL mode bit set, so such a value
 *                                   saturates (hi = +-65504, lo = f16-clamped remainder: finite, never inf / NaN) instead of
 *                                   poisoning the sum; nlc_conv_desc.debug bit 1 turns it into an error (NLC_EINVAL). */
enum { NLC_MATH_NATIVE = 0, NLC_MATH_F16X3 = 1 };
enum { NLC_OK = 0, NLC_EINVAL = -1, NLC_ELAUNCH = -2, NLC_EUNSUPPORTED = -3 };
enum { NLC_ACT_NONE = 0, NLC_ACT_SILU = 1, NLC_ACT_GELU = 2 };
enum { NLC_OUT_NHWC = 0, NLC_OUT_NCHW_F32 = 1 };
/* scheduler variants of nlc_sched_step (reference src/schedulers.py:425-627) */
enum {
    NLC_SCHED_DDIM = 0,            /* DDIM_Scheduler.pred_xprev             :432-449 */
    NLC_SCHED_DDIM_SIMPLE = 1,     /* DDIM_simple_Scheduler.pred_xprev      :465-473 */
    NLC_SCHED_DDIM_SIMPLE_ORIG = 2,/* DDIM_simple_orig_Scheduler.pred_xprev :487-496 */
    NLC_SCHED_DDIM_SIMPLE_DRAG = 3,/* DDIM_simple_drag_Scheduler.pred_xprev :505-514 */
    NLC_SCHED_DDPM = 4,            /* DDPM_Scheduler.pred_xprev             :548-562 */
    NLC_SCHED_DDPM_ORIG = 5,       /* DDPM_orig_Scheduler.pred_xprev        :581-599 */
    NLC_SCHED_DDIM_ORIG = 6        /* DDIM_orig_Scheduler.pred_xprev        :609-627 */
};
enum { NLC_CLIP_NONE = 0, NLC_CLIP_CLAMP = 1, NLC_CLIP_DYNAMIC = 2 };
enum { NLC_VAR_NONE = 0, NLC_VAR_FIXEDSMALL = 1, NLC_VAR_FIXEDLARGE = 2, NLC_VAR_LEARNED = 3 };
/* kernel-selection policy of one nlc_conv2d call (nlc_conv_desc.policy).  AUTO is the production dispatch: the
 * LDS-halo kernel for stride-1 3x3 "same" convolutions with >= 128 (16x16 pixel x 128 channel) tiles, else the LDS-DMA
 * implicit-GEMM kernel (3x3 / 1x1), else the generic gather kernel.  The others exist so that parity tests and A/B
 * timings can pin a kernel per call; there is no process-wide switch. */
enum { NLC_CONV_AUTO = 0, NLC_CONV_FORCE_HALO = 1, NLC_CONV_NO_HALO = 2, NLC_CONV_GENERIC = 3,
       NLC_CONV_FORCE_SMALL = 4 }; /* the small-map 3x3 kernel (conv_small.hip) for every shape it takes, with or without gn_in */

int nlc_version(void);
const char* nlc_last_error(void);

/* Tile geometry the host needs to pack weights: Cout is padded to a multiple of
 * *cout_mult rows and Cin to a multiple of *cin_mult elements (zero filled). */
int nlc_conv_pack_dims(int dtype, int* cout_mult, int* cin_mult);

/* Pack one convolution / linear weight for nlc_conv2d, on the device (load time, once per tensor):
 *   w        : the reference's own layout, f32 [Cout][Cin][KH*KW]  (nn.Conv2d / Conv1d(k=1) / Linear .weight, contiguous)
 *   packed   : [Cout_pad][KH*KW][Cin_pad] in the compute dtype, zero padded; Cout_pad / Cin_pad = Cout / Cin rounded up to
 *              the multiples nlc_conv_pack_dims reports
 *   packed[r][tap][c] = (T)( (double) w[row_perm ? row_perm[r] : r][col_perm ? col_perm[c] : c][tap] * (row_scale ? row_scale[r] : 1) )
 *   bias_out[r]       = (float)( (double) bias[row_perm ? row_perm[r] : r] * (row_scale ? row_scale[r] : 1) + (bias_add ? bias_add[r] : 0) )
 * which covers every load-time fold of this path:
 *   row_perm   int32 [Cout]   output-channel order: the reference's qkv layouts -> the canonical [q|k|v][head][ch]
 *                             (legacy [head][q|k|v][ch], src/unet_adm.py:347; new order :380-388)
 *   row_scale  f64   [Cout]   per-output-channel factor: the attention scale ch^-1/4 on q and k (src/unet_adm.py:348-351),
 *                             1/sqrt(C) on k (src/edm_networks.py:127), BatchNorm1d(eval) gamma/sqrt(var+eps)
 *                             (src/unet_adm.py:1056)
 *   bias_add   f64   [Cout]   BatchNorm's beta - mean*gamma/sqrt(var+eps)
 *   col_perm   int32 [Cin]    input-feature order: NCHW-flatten -> NHWC-flatten of the sigma head's Linear (:1055)
 * bias may be NULL (then bias_out is written only if bias_add is given; a NULL bias counts as 0); bias_out may be NULL
 * when neither exists.  All pointers are device pointers. */
int nlc_pack_conv_weights(const float* w, const float* bias, int Cout, int Cin, int KH, int KW,
                          const int32_t* row_perm, const double* row_scale, const double* bias_add,
                          const int32_t* col_perm, int dtype, void* packed, float* bias_out, void* stream);
/* same with the matrix-arithmetic mode the weights will be used with (nlc_conv_desc.math).  math = NLC_MATH_F16X3 (dtype must
 * be NLC_F32): every 32-channel k-block (128 bytes) of a packed row holds, per 16-byte chunk c = 0..3, the f16 `hi` halves of
 * channels 4c..4c+3 and 16+4c..16+4c+3 (the 8 k-values one lane feeds to a 16x16x32 MFMA), and chunk 4+c the `lo` halves of the
 * same channels; hi = f16(x) rounded to nearest, lo = f16(x - hi), where x = 2^e[r] * (the value packed[r][tap][c] above) and
 * 2^e[r] is the power of two that brings max |row r| into [2^14, 2^15) (e = 0 for an all-zero or padding row).
 *   w_scale_out  f32 [Cout_pad], REQUIRED for NLC_MATH_F16X3 (ignored, may be NULL, otherwise): w_scale_out[r] = 2^-e[r];
 *                hand it to nlc_conv2d as nlc_conv_desc.w_scale.  bias_out is NOT scaled. */
int nlc_pack_conv_weights_ex(const float* w, const float* bias, int Cout, int Cin, int KH, int KW,
                             const int32_t* row_perm, const double* row_scale, const double* bias_add,
                             const int32_t* col_perm, int dtype, int math, void* packed, float* bias_out, float* w_scale_out,
                             void* stream);

/* ------------------------------------------------------------------------------------
 * Convolution / linear as one implicit-GEMM MFMA kernel.
 * Replaces torch.nn.functional.conv2d / conv1d(k=1) / linear at
 *   src/unet_adm.py:98,131-133,185,211,222,286,294,617 ; src/unet_simple.py:41,61,87,96,109,143-162,226,296
 *   src/edm_networks.py:42,95 ; nn.Linear at src/unet_adm.py:474-476,201,1055,1058
 * together with the elementwise work the reference does right after it:
 *   + bias, + per-(image,channel) embedding add (unet_simple.py:121, edm_networks.py:192),
 *   + residual add (unet_adm.py:256,305), * out_scale (edm_networks.py:196), activation.
 * Input may be the channel-concatenation of two NHWC tensors (th.cat at unet_adm.py:662)
 * and may be read through a nearest-neighbour 2x upsample (F.interpolate, unet_adm.py:107).
 * ---------------------------------------------------------------------------------- */
/* GroupNorm (+FiLM) (+SiLU) of a convolution's INPUT, described by the ride-along totals of the launches that produced the input
 * (nlc_conv_desc.gn_in, ABI v7): the small-map 3x3 kernel normalises its input slice on the way into LDS, so the GroupNorm + SiLU in
 * front of a ResBlock's convolutions (/root/reference/src/unet_adm.py:182-185,206-211,248-252; src/unet_simple.py:117-124;
 * src/edm_networks.py:185-192) costs neither a launch nor a pass over HBM.  Same arithmetic as nlc_groupnorm_prestats:
 *   y = act( ((x - mean) * rstd * gamma + beta) * (1 + scale[b][c]) + shift[b][c] ),   zero padding stays zero.
 * Fields as in nlc_groupnorm_prestats: stats0 / stats1 = int64 [B][C0/g0 | C1/g1][4] totals of x0 / x1 (stats1 NULL iff C1 == 0),
 * granule 0 | 8 | 4; (C0+C1)/groups a multiple of both granules; gamma / beta f32 [C0+C1] or NULL; scale / shift f32 row-strided
 * [B][>= C0+C1] views (both or neither); act = NLC_ACT_NONE | NLC_ACT_SILU. */
typedef struct nlc_gn_in {
    const void* stats0; const void* stats1;
    int32_t granule0, granule1;
    int32_t groups;
    float eps;
    const float* gamma; const float* beta;
    const float* scale; const float* shift;
    int32_t ss_stride;
    int32_t act;
} nlc_gn_in;

typedef struct nlc_conv_desc {
    const void* x0;      /* [B][Hin][Win][C0]                                   */
    const void* x1;      /* [B][Hin][Win][C1] or NULL; logical input = cat(x0,x1)*/
    int32_t C0, C1;
    int32_t B, Hin, Win; /* physical input size (before the optional 2x upsample) */
    int32_t Hout, Wout;
    int32_t Cout;        /* real output channels                                 */
    int32_t KH, KW, stride, pad_t, pad_l;
    int32_t upsample2x;  /* 1: taps index a virtual (2Hin x 2Win) nearest-upsampled input */
    const void* w;       /* packed [Cout_pad][KH*KW][Cin_pad], compute dtype      */
    int32_t Cin_pad, Cout_pad;
    const float* bias;   /* [Cout] f32 or NULL                                   */
    const float* emb;    /* [B][emb_stride] f32 or NULL: out[b,:,:,n] += emb[b][n] */
    int32_t emb_stride;
    const void* res;     /* [B][Hout][Wout][Cout] compute dtype or NULL          */
    float out_scale;     /* applied after bias/emb/res, before act               */
    int32_t act;         /* NLC_ACT_*                                            */
    void* out;           /* NLC_OUT_NHWC: compute dtype; NLC_OUT_NCHW_F32: float */
    int32_t out_mode;
    void* workspace;     /* optional scratch (NULL = none): enables split-K on shapes with few output tiles and a long */
    int64_t workspace_bytes; /* K (the 8x8 / 16x16 levels); size from nlc_conv2d_workspace_bytes.  Its FIRST 4096 BYTES are    */
                         /* the library's arrival counters: zero them once when the buffer is allocated and never write them; */
                         /* every launch leaves them zero.  The rest is undefined scratch.  One workspace per stream.          */
    void* stats_out;     /* optional (NULL = none): GroupNorm statistics of the output ride along in the conv's epilogue so   */
    int64_t stats_bytes; /* that the normalisation that follows (src/unet_adm.py:182-184,206-208) needs NO pass over the     */
                         /* tensor, no reduction and no finalize launch: int64 [B][Cout/g][4] = per g-channel chunk the TOTALS */
                         /* (sum.hi, sum.lo, sumsq.hi, sumsq.lo) of the STORED (rounded) values, g = stats_granule.  Every      */
                         /* workgroup ADDS its contributions with 64-bit integer atomics - a contribution v (an f32 partial    */
                         /* sum) as hi = floor(v), lo = floor((v - hi) 2^44); value = hi + lo 2^-44 - so the totals are exact   */
                         /* sums of the contributions in ANY arrival order: bit-reproducible.  The CALLER ZEROES the buffer    */
                         /* before the launch (the library only adds); 16-byte aligned (checked).  A contribution that is inf, */
                         /* NaN or >= 2^45 in magnitude sets bit 62 of its chunk's sumsq.hi word instead (atomic OR; no add      */
                         /* reaches or clears that bit): the consumers then give that chunk's groups NaN (mean, rstd), as         */
                         /* F.group_norm does for a group holding a non-finite value.  nlc_conv2d_stats_partials(desc,          */
                         /* dtype) must be > 0 (16-bit, NHWC, Cout % 128 == 0, the LDS-halo and LDS-DMA kernels).  Consumed by */
                         /* nlc_groupnorm_prestats / _pool2x2 / _coef.                                                        */
    int32_t policy;      /* NLC_CONV_* (0 = AUTO); the three queries below honour it like nlc_conv2d does */
    const float* gn_coef; /* optional (NULL = none): the GroupNorm (+FiLM) (+SiLU) that precedes this convolution in the reference */
    int32_t gn_act;      /* (src/unet_adm.py:182-185,206-211,248-252) applied to the INPUT on its way through LDS instead of in a     */
                         /* separate pass over HBM: logical input = act(a[b][c] * x + b[b][c]) inside the image, 0 in the padding;    */
                         /* gn_coef = float [B][C0+C1][2] = (a, b) from nlc_groupnorm_coef, allocated with >= 512 bytes of slack      */
                         /* behind it; gn_act = NLC_ACT_NONE | NLC_ACT_SILU.  Only launches for which                                 */
                         /* nlc_conv2d_prologue_supported(desc, dtype) returns 1 take it (bf16, LDS-halo kernel).                     */
    int32_t tuning;      /* 0 in production.  Bit mask of kernel A/B switches for in-process timing experiments (tools/): */
                         /* results are identical for every value, only the schedule changes (see conv_halo.hip).          */
    int32_t res_upsample2x; /* 1: res is [B][Hout/2][Wout/2][Cout] and output pixel (y, x) adds res pixel (y/2, x/2) - the        */
                         /* skip branch x_upd(x) of an up-sampling ResBlock (src/unet_adm.py:186-190, nearest-2x Upsample) read    */
                         /* in place of a materialised upsampled copy.  Hout, Wout must be even.                                  */
    int32_t stats_granule; /* channels per chunk of stats_out: 0 or 8 = per 8 channels, 4 = per 4 (for a GroupNorm whose groups are 4 or 12 ...  */
                         /* channels wide: 128 channels in 32 groups); stats_out is then int64 [B][Cout/4][4].                          */
    int32_t math;        /* NLC_MATH_* (0 = the dtype's native MFMA).  NLC_MATH_F16X3 needs dtype NLC_F32 and `w` packed by          */
                         /* nlc_pack_conv_weights_ex(..., NLC_MATH_F16X3): the packed tensor then holds (hi, lo) f16 halves of every  */
                         /* weight in the k order the kernels read, same byte size as the f32 packing.                                 */
    int32_t debug;       /* 0 in production.  Bit 0: before launching a split-K shape, copy the workspace's arrival counters back and   */
                         /* return NLC_EINVAL if any is non-zero (a poisoned workspace; synchronises the stream - tests / triage only). */
                         /* Bit 1 (NLC_MATH_F16X3): reduce max |x| over the input tensor(s) first and return NLC_EINVAL if it is >= 65504 */
                         /* or not finite - outside the domain of the operand split (synchronises the stream - tests / triage only).     */
    const float* w_scale; /* NLC_MATH_F16X3: f32 [Cout_pad] from nlc_pack_conv_weights_ex (w_scale_out), REQUIRED in that mode: the     */
                         /* conv sum of output channel n is multiplied by w_scale[n] (a power of two: exact) before bias / embedding /  */
                         /* residual are added.  Must be NULL for NLC_MATH_NATIVE.                                                       */
    void* norm_out;      /* NULL, or (ABI v6) [B][Hin][Win][C0+C1] of the tensor dtype: a pointwise (1x1, stride 1) launch then ALSO writes */
                         /* gn_act(a[b][c] * x + b[b][c]) of its input there (gn_coef REQUIRED: the table of nlc_groupnorm_coef), while    */
                         /* the convolution itself runs on x as given - the skip projection of a ResBlock and the GroupNorm + SiLU in      */
                         /* front of its first 3x3 from ONE read of cat(x0, x1) (/root/reference/src/unet_adm.py:236-256: in_layers(x)      */
                         /* and skip_connection(x)).  Only launches for which nlc_conv2d_norm_out_supported(desc, dtype) returns 1          */
                         /* (16-bit, C0 and C1 multiples of 128, C0+C1 <= 512, maps of whole multiples of 64 pixels, >= 1536 output tiles). */
    const nlc_gn_in* gn_in; /* NULL, or (ABI v7) the normalisation of the input, applied by the convolution itself on the input's way into  */
                         /* LDS (see nlc_gn_in).  Only launches for which nlc_conv2d_gn_in_supported(desc, dtype) returns 1: 16-bit 3x3 /    */
                         /* stride 1 / pad 1 on 8-, 16- or 32-pixel-wide maps, whole 128-pixel x 128-channel tiles, C0 and C0+C1 multiples */
                         /* of 64.  Split-K needs the workspace of nlc_conv2d_workspace_bytes (same layout as for the other kernels).   */
                         /* With 2 / 4 / 8 splits the tile's workgroups WAIT for each other (distributed reduction): the launch has at   */
                         /* most one workgroup per CU and expects to be resident at once - true on a stream of its own kernels; a caller */
                         /* that runs other long-lived kernels CONCURRENTLY on other streams sets tuning bit 23 (the last arriver then   */
                         /* reduces alone, nobody waits; same bits).  The wait is bounded (~0.2 s) and leaves a non-zero mark in the     */
                         /* workspace's counter word 1023 when it gives up (reported by debug bit 0).                                   */
} nlc_conv_desc;

int nlc_conv2d(const nlc_conv_desc* d, int dtype, void* stream);
/* bytes of workspace with which nlc_conv2d would split K for this descriptor (0: it would not).  Passing the workspace
 * is the caller's opt-in: with none, one summation order is kept (the f32 parity path).  Partial sums are f32 and are
 * added in a fixed order. */
int64_t nlc_conv2d_workspace_bytes(const nlc_conv_desc* d, int dtype);
/* 1 if nlc_conv2d would add this launch's GroupNorm statistics to stats_out for this descriptor, 0: it would not */
int nlc_conv2d_stats_partials(const nlc_conv_desc* d, int dtype);
/* 1 if nlc_conv2d would apply desc.gn_coef / gn_act in its LDS prologue for this descriptor (geometry, dtype and policy decide;
 * the gn_* fields themselves are not looked at), else 0: the caller then runs nlc_groupnorm(_prestats) as a separate pass */
int nlc_conv2d_prologue_supported(const nlc_conv_desc* d, int dtype);
/* 1 if nlc_conv2d would honour desc->norm_out for this descriptor (norm_out / gn_coef themselves are not looked at), else 0 */
int nlc_conv2d_norm_out_supported(const nlc_conv_desc* d, int dtype);
/* 1 if nlc_conv2d would honour desc->gn_in for this descriptor (the small-map 3x3 kernel takes it; gn_in itself is not looked at),
 * else 0: the caller then runs nlc_groupnorm(_prestats) as a separate pass.  nlc_conv2d_workspace_bytes / _stats_partials answer for
 * the kernel that WILL run: set desc->gn_in (or policy NLC_CONV_FORCE_SMALL) before asking them. */
int nlc_conv2d_gn_in_supported(const nlc_conv_desc* d, int dtype);

/* EXPERIMENT, measured and not used by the networks ("One launch per ResBlock"): both 3x3 convolutions of a ResBlock
 * whose input and output widths are equal (/root/reference/src/unet_adm.py:236-256 without a skip projection) in ONE launch of the
 * small-map kernel - conv1 (d1, with d1->gn_in = in_layers' GroupNorm + SiLU), a grid-wide barrier, conv2 (d2, with d2->gn_in =
 * out_layers' GroupNorm (+FiLM) + SiLU described by d1->stats_out, d2->x0 = d1->out, d2->res = the block input).  Both descriptors
 * must be small-map launches that tile alike (nlc_conv2d_gn_in_supported), share ONE workspace, and their grid must be resident
 * at once (at most one workgroup per CU) - else NLC_EUNSUPPORTED and nothing is launched.  barrier: two ints, zero between launches
 * (the kernel leaves them zero).  Results are those of nlc_conv2d(d1) followed by nlc_conv2d(d2), bit for bit. */
int nlc_resblock_small(const nlc_conv_desc* d1, const nlc_conv_desc* d2, void* barrier, int dtype, void* stream);

/* First-layer convolution for tiny Cin (<=4): reads the sampler state in the reference's
 * own layout (NCHW f32), applies the per-sample input scale c_in[b] (convert_coordinate,
 * src/experiments.py:273-282; c_in of :782,796), and writes NHWC compute dtype.
 * Replaces conv_nd at src/unet_adm.py:484, conv_in at src/unet_simple.py:226,
 * enc '<res>_conv' at src/edm_networks.py:790.   w: [Cout][KH*KW][Cin] f32. */
int nlc_conv_first(const float* x_nchw, const float* in_scale /*[B] or NULL*/,
                   const float* w, const float* bias, void* out_nhwc,
                   int B, int Cin, int H, int W, int Cout, int KH, int KW,
                   int dtype, void* stats_out /* or NULL */, int64_t stats_bytes, int stats_granule /* 0 | 8 | 4 */, void* stream);
/* stats_out: GroupNorm statistics of the output, int64 [B][Cout/granule][4] totals as in nlc_conv_desc.stats_out (zeroed by the
 * caller, added to by the kernel), when nlc_conv_first_stats_partials(...) > 0 (16-bit, KH*KW*Cin <= 32, Cout <= 256 and a
 * multiple of 16, H*W a multiple of 64: the launches that take the matrix-core kernel). */
int nlc_conv_first_stats_partials(int Cin, int H, int W, int Cout, int KH, int KW, int dtype);

/* ------------------------------------------------------------------------------------
 * GroupNorm (+ optional FiLM scale/shift, + optional SiLU), statistics in f32.
 * Replaces GroupNorm32 + SiLU (src/nn_util.py:17-19, src/unet_adm.py:182-184,206-208,
 * 248-252), Normalize+nonlinearity (src/unet_simple.py:32,117-118,123-124) and
 * GroupNorm+silu (src/edm_networks.py:113-116,185,192).  Input may be cat(x0,x1).
 *   y = act( ((x-mean)*rstd*gamma+beta) * (1+scale[b,c]) + shift[b,c] )
 * workspace: at least nlc_groupnorm_workspace_bytes() bytes, f32 aligned.
 * ---------------------------------------------------------------------------------- */
int64_t nlc_groupnorm_workspace_bytes(int B, int HW, int C, int groups);
int nlc_groupnorm(const void* x0, const void* x1, int C0, int C1, int B, int HW,
                  int groups, float eps, const float* gamma, const float* beta,
                  const float* scale, const float* shift, int ss_stride,
                  int silu, void* out, void* workspace, int dtype, void* stream);
/* Same, but the statistics come from the producing convolutions' epilogues (nlc_conv_desc.stats_out) instead of a pass
 * over the input: stats0 / stats1 = the int64 [B][C0/g0 | C1/g1][4] totals of x0 / x1.  ONE launch: every thread of the
 * streaming kernel derives its channels' coefficients from the totals of the one or two groups they lie in (f64:
 * var = E[x^2] - E[x]^2).  Requires a 16-bit dtype, (C0+C1)/groups a multiple of both granules and C0, C1 multiples of 8 (a
 * group may straddle the two sources).  No workspace. */
int nlc_groupnorm_prestats(const void* x0, const void* x1, int C0, int C1, int B, int HW,
                           int groups, float eps, const float* gamma, const float* beta,
                           const float* scale, const float* shift, int ss_stride,
                           int silu, void* out, int dtype,
                           const void* stats0, int granule0, const void* stats1, int granule1, void* stream);
/* granule0 / granule1: channels per chunk of stats0 / stats1 (0 or 8, or 4 - nlc_conv_desc.stats_granule of the producer); the
 * group size must be a multiple of both. */

/* The two branches of a down-sampling ResBlock from ONE read of x (src/unet_adm.py:193-195: h = in_rest(x); h = h_upd(h);
 * x = x_upd(x), both AvgPool2d(2)):  out_norm = avgpool2x2(act(GroupNorm(x) (FiLM))),  out_x = avgpool2x2(x), both
 * [B][H/2][W/2][C]; the full-resolution normalised tensor is never written.  Statistics: stats0 as in
 * nlc_groupnorm_prestats (16-bit), or NULL -> computed here by a pass over x.  f32: bit-identical to nlc_groupnorm followed by
 * nlc_avgpool2x2; 16-bit: the pooled mean is taken of the f32 activations (one rounding less).  Same workspace as nlc_groupnorm
 * (may be NULL when stats0 is given). */
int nlc_groupnorm_pool2x2(const void* x, int C, int B, int H, int W, int groups, float eps, const float* gamma,
                          const float* beta, const float* scale, const float* shift, int ss_stride, int silu,
                          void* out_norm, void* out_x, void* workspace, int dtype, const void* stats0, int granule0,
                          void* stream);

/* The per-(image, channel) coefficients of the same normalisation, for the convolution that applies it in its LDS prologue
 * (nlc_conv_desc.gn_coef): coef[b][c] = (a, b) with  a = rstd*gamma*(1+scale),  b = (beta - mean*rstd*gamma)*(1+scale) + shift,
 * statistics from the producing convolutions' epilogues exactly as in nlc_groupnorm_prestats (same f64 arithmetic on the totals).
 * coef: float [B][C0+C1][2] (+ >= 512 bytes of slack behind it for the consumer's DMA). */
int nlc_groupnorm_coef(int C0, int C1, int B, int HW, int groups, float eps, const float* gamma, const float* beta,
                       const float* scale, const float* shift, int ss_stride, const void* stats0, int granule0,
                       const void* stats1, int granule1, float* coef, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-head softmax attention on token-major tensors (flash style, no TxT matrix in HBM).
 * Replaces QKVAttentionLegacy/QKVAttention.forward (src/unet_adm.py:337-354,370-389),
 * AttnBlock bmm/softmax/bmm (src/unet_simple.py:172-185) and AttentionOp + einsum
 * (src/edm_networks.py:126-130,199-203).  The per-family q/k scaling is folded into the
 * qkv projection weights by the host, so here  out = softmax(q k^T) v  exactly.
 *   qkv: [B][T][3][H][D]   out: [B][T][H][D]   (compute dtype), softmax in f32.
 *   logit_log2: 0 = q k^T are natural logits (e^x); 1 = the host ALSO folded log2(e) into the q (or k) projection - one more
 *   factor in the same f64 row scale, no extra rounding - so the logits arrive in log2 units and out = softmax_2(q k^T) v with 2^x
 *   in place of e^x: the same function of the network's weights, one multiply per score less in the kernel.
 * ---------------------------------------------------------------------------------- */
int nlc_attention(const void* qkv, void* out, int B, int T, int H, int D,
                  int dtype, int logit_log2, void* stream);

/* ---- resampling on NHWC (src/unet_adm.py:107,136 ; src/unet_simple.py:48,73 ;
 *      edm_networks.py:88-93 with resample_filter [1,1]) ---- */
int nlc_avgpool2x2(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream);
int nlc_upsample2x(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream);
/* zero-pad right/bottom by one pixel (F.pad (0,1,0,1), src/unet_simple.py:69-70) */
int nlc_pad_rb(const void* x, void* out, int B, int H, int W, int C, int dtype, void* stream);
/* NHWC compute dtype  ->  NCHW f32 (feature map handed to the sigma net's callers) */
int nlc_nhwc_to_nchw_f32(const void* x, float* out, int B, int H, int W, int C, int dtype, void* stream);
/* NCHW f32 -> NHWC compute dtype */
int nlc_nchw_f32_to_nhwc(const float* x, void* out, int B, int H, int W, int C, int dtype, void* stream);

/* ---- sinusoidal timestep embeddings, f32 out [B][dim]: out = [cos(t*f) || sin(t*f)] or, with
 *  sin_first, [sin || cos].  freqs[dim/2] is computed on the host with the reference's own
 *  formula so that t*f is bit-identical:
 *    ADM    src/nn_util.py:103-121      cos||sin, f = exp(-ln(1e4) i/half)
 *    simple src/unet_simple.py:6-24     sin||cos, f = exp(-i ln(1e4)/(half-1))
 *    EDM    src/edm_networks.py:219-225 + swap at :838 -> sin||cos, f = (1e-4)^(i/(half-1)) ---- */
int nlc_timestep_embedding(const float* t, const float* freqs, float* out, int B, int dim,
                           int sin_first, void* stream);

/* ------------------------------------------------------------------------------------
 * Sampler-state kernels (per-sample, f32, NCHW state exactly as the reference keeps it).
 * The per-sample scalars sigma_t[b], sigma_prev[b], t[b], c_in[b] live in device arrays.
 * ---------------------------------------------------------------------------------- */
/* sumsq[b] = sum_{d<D} x[b*row_stride+d]^2  (utils.vector_norm squared, src/utils.py:7-9).
 * row_stride > D selects the first D elements of each row (the eps channels of a 2C-channel
 * network output, src/experiments.py:451-458). */
int nlc_row_sumsq(const float* x, float* sumsq, int B, int64_t row_stride, int64_t D, void* stream);

/* refine_prior_sigma + timestep lookup (src/experiments.py:401-419):
 *   refine: n = sqrt(sumsq[b])/sqrt_dim ; sigma_t = clamp(sigma_sched, max(n-norm_max,0), n+norm_min)
 *           t = searchsorted_left(sigmas, sigma_t) ; if min_b t > 0: t -= time_shift
 *   else  : sigma_t = sigma_sched, t = t_sched
 *   always: sigma_prev = sigma_prev_sched ; t = clamp(t,0,1000) (f32) ; c_in = sqrt(1/(sigma_t^2+1))
 *   norm_max / norm_min are already divided by sqrt(dim) (set_norm_maxmin, :176-184).
 *   sigmas: the scheduler's ascending table (src/schedulers.py:134). */
int nlc_refine_sigma(const float* sumsq, float sqrt_dim, float norm_max, float norm_min,
                     float sigma_sched, float sigma_prev_sched, int refine,
                     const float* sigmas, int n_sigmas, int t_sched, int time_shift,
                     float* sigma_t, float* sigma_prev, float* t, float* c_in,
                     int B, void* stream);

/* General form of nlc_refine_sigma: per-sample scheduled inputs (projection_loop carries sigma_t and t per
 * sample from step to step, image_sample.py:461-497) and continuous-t schedules.
 *   sigma_in / t_in   : [B] per-sample scheduled values, or NULL -> the scalars sigma_sched / t_sched.  They may
 *                       alias sigma_t / t (each sample is read before it is written).
 *   prev_is_ratio     : sigma_prev[b] = raw_sigma[b] * sigma_prev_sched (recal_sigma_prev, image_sample.py:463-464)
 *                       instead of the scalar sigma_prev_sched; raw_sigma is the value BEFORE the refine clamp.
 *   t_slopes          : NULL -> discrete t = searchsorted_left(sigmas, sigma) (src/schedulers.py:185-190);
 *                       else [n_sigmas-1] f32 slopes 1/(eps+sigmas[i+1]-sigmas[i]) -> continuous t by linear
 *                       interpolation, t = i + slopes[i]*(sigma - sigmas[i]), i = clamp(searchsorted-1, 0, n-2)
 *                       (sigma_to_t_interp, src/schedulers.py:210-220; src/torchinterp1d.py:10-154).
 *   time_shift        : subtracted from every t when refine and min_b t > 0 (src/experiments.py:411-412). */
typedef struct nlc_sigma_desc {
    const float* sumsq;      /* [B] row sums of squares of xt (refine only) */
    const float* sigma_in;   /* [B] or NULL */
    const float* t_in;       /* [B] or NULL */
    const float* sigmas;     /* [n_sigmas] ascending table */
    const float* t_slopes;   /* [n_sigmas-1] or NULL */
    float* sigma_t;          /* out [B] */
    float* sigma_prev;       /* out [B] */
    float* t;                /* out [B] */
    float* c_in;             /* out [B] */
    float sqrt_dim, norm_max, norm_min;
    float sigma_sched, sigma_prev_sched, t_sched, time_shift;
    int refine, prev_is_ratio, n_sigmas, B;
} nlc_sigma_desc;
int nlc_refine_sigma_ex(const nlc_sigma_desc* d, void* stream);

/* NLC correction (src/experiments.py:424-431): sigma_hat = sigma_t*(1+r[b]),
 * sigma_prev_hat = sigma_hat*(sigma_prev/sigma_t) (skipped when partial != 0, style
 * 'pred_partial'), t = clamp(lookup(sigma_hat),0,1000) with the lookup of nlc_sigma_desc.t_slopes
 * (NULL = discrete searchsorted_left), c_in updated. */
int nlc_sigma_correct(const float* r, int partial, const float* sigmas, const float* t_slopes, int n_sigmas,
                      float* sigma_t, float* sigma_prev, float* t, float* c_in,
                      int B, void* stream);

/* projection_loop's per-step sigma re-estimation (image_sample.py:485-497), in place on the per-sample state:
 *   cur_norm = sqrt(sumsq[b])/sqrt_dim                       (sumsq of the NEW x_{t-1})
 *   cur_dist = sqrt(cur_norm^2 + norm_max_sq - 2*cur_norm*norm_max*costheta + 1e-8)
 *   sigma_t[b] <- term0 + r1*sigma_prev[b] + r2*sigma_t[b]*(cur_norm/last_norm[b]) + r3*cur_dist
 *   t[b] <- lookup(sigma_t[b]) (not clamped here; the next step clamps) ; last_norm[b] <- cur_norm
 * term0 = sigma_estimate_rate[0] * scheduled sigma_prev (a host scalar). */
int nlc_proj_sigma(const float* sumsq, float sqrt_dim, float norm_max, float norm_max_sq, float costheta,
                   float term0, float r1, float r2, float r3, const float* sigmas, const float* t_slopes,
                   int n_sigmas, float* last_norm, float* sigma_t, const float* sigma_prev, float* t,
                   int B, void* stream);

/* s[b] = clamp(quantile_q(|x[b,:]|), 1, max_value) with torch.quantile's linear
 * interpolation and f32 rank arithmetic (src/experiments.py:190-199), exact radix select. */
int nlc_dynamic_threshold(const float* x0_hat, float q, float max_value, float* s_out,
                          int B, int64_t D, void* stream);
/* The same result (bit-identical) with G workgroups per sample and one launch pair per digit pass - the form the sampling loop uses:
 * the single-workgroup kernel above keeps B CUs busy.  `workspace`: nlc_dynamic_threshold_ws_bytes(B) bytes of device memory that
 * the caller zeroes ONCE; every call finds it zero and leaves it zero (calls on one workspace must be stream-ordered). */
int64_t nlc_dynamic_threshold_ws_bytes(int B);
int nlc_dynamic_threshold_ws(const float* x0_hat, float q, float max_value, float* s_out, int B, int64_t D,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* One fused scheduler update on the NCHW f32 state (src/experiments.py:360-370,
 * src/schedulers.py:367-390,407-449 and the variants listed in the NLC_SCHED_* enum).
 *   eps_out: network output [B][Cnet][H][W]; channels [0,C) are eps, [C,2C) the learned
 *            log-variance fraction when var_mode == NLC_VAR_LEARNED (learn_sigma).
 *   eps_norm_sumsq: if non-NULL, eps <- sqrt(D)*eps/max(sqrt(sumsq[b]),1e-12)
 *                   (utils.normalize, src/utils.py:11-16).
 *   nlc_sched_x0  : x0 <- xt - sigma_t*eps                     (pred_xstart, pre-clip)
 *   nlc_sched_step: x0 <- clip(x0) [none | clamp(-1,1) | clamp(-s,s)/s with s = dyn_s[b]],
 *                   optional inpainting projection x0 <- known where mask != 0,
 *                   x_prev <- pred_xprev(x0, eps, ...) for the chosen variant.
 *   noise: host-ordered N(0,1) draw [B][C][H][W] (needed when eta > 0 or DDPM variants).
 *   nan_flag: optional device int, OR-ed with 1 if any x_prev is NaN (the reference's
 *             torch.isnan(xt).any() early break, src/experiments.py:389).
 */
typedef struct nlc_sched_desc {
    const float* xt; const float* eps_out; const float* noise;
    const float* known; const float* mask;           /* [B][C][HW] / [C][HW], or both NULL */
    const float* sigma_t; const float* sigma_prev;   /* [B] */
    const float* eps_norm_sumsq;  /* [B] or NULL */
    const float* dyn_s;           /* [B] or NULL (dynamic threshold value) */
    const float* logvar_ext;      /* [B][C][HW] caller-supplied log-variance (overrides var_mode) or NULL */
    float* x0; float* x_prev; float* eps_used;       /* eps_used may be NULL */
    int32_t B, C, Cnet, HW;
    int32_t variant, clip, var_mode;
    int32_t phases;               /* bit 0: clip (+mask) x0 in place ; bit 1: write x_prev.  0 means both */
    float eta, min_var_coef;
} nlc_sched_desc;
int nlc_sched_x0(const nlc_sched_desc* d, void* stream);
int nlc_sched_step(const nlc_sched_desc* d, int* nan_flag, void* stream);

/* out[b,:] = x[b,:] * scale[b] * scalar  (scale may be NULL).  inv_convert_coordinate,
 * src/experiments.py:284-293: xt = z * sqrt(sigma0^2+1). */
int nlc_scale_rows(const float* x, const float* scale /*[B]*/, float scalar, float* out,
                   int B, int64_t D, void* stream);

/* out[b,:] = ca[b]*x[b,:] + cb[b]*y[b,:]  (y, cb may both be NULL), products and sum rounded separately.  The training-time
 * forward process  x_n = x_0 sqrt(abar_t) + noise sqrt(1 - abar_t)  (Scheduler.diffusion, src/schedulers.py:323-329) that
 * feeds the frozen-encoder feature extraction of the sigma-net training loop (src/experiments.py:656-681). */
int nlc_lincomb_rows(const float* x, const float* ca /*[B]*/, const float* y, const float* cb /*[B]*/, float* out,
                     int B, int64_t D, void* stream);

/* ------------------------------------------------------------------------------------
 * EDM / Heun + NLC sampler state (src/experiments.py:777-918): the state and eps are FLOAT64,
 * the network runs in float32, exactly as in the reference (:860,872,789-802).
 * ---------------------------------------------------------------------------------- */
/* out[i] = (float)x[i]                       xt.to(torch.float32), :778,789 */
int nlc_cast_f64_f32(const double* x, float* out, int64_t n, void* stream);
/* sumsq[b] = sum_d x[b,d]^2 in f64            vector_norm on the f64 state, :808,842 */
int nlc_row_sumsq_f64(const double* x, double* sumsq, int B, int64_t D, void* stream);

/* out[b] = cosine_similarity(a[b,:], b[b,:]) = sum_d (a/max(||a||,eps)) * (b/max(||b||,eps)) in f64:
 * torch.nn.CosineSimilarity(dim=1, eps=1e-6) of the Heun step's eps_scale=None branch
 * (src/experiments.py:870,912-915). */
int nlc_row_cosine_f64(const double* a, const double* b, double eps, double* out, int B, int64_t D, void* stream);
/* preconditioning scalars from sigma[b] cast to f32 (:790-797):
 *   c_skip = sd^2/(s^2+sd^2), c_out = s*sd/sqrt(s^2+sd^2), c_in = 1/sqrt(sd^2+s^2), c_noise = ln(s)/4 */
int nlc_edm_scalars(const double* sigma, float sigma_data, float* c_in, float* c_noise,
                    float* c_skip, float* c_out, int B, void* stream);
/* D_x = c_skip*x32 + c_out*F (f32) ; denoised = (double)D_x ; eps = (x - denoised)/sigma_div[b]  (:801,836-840)
 * denoised may be NULL. */
int nlc_edm_eps(const double* x, const float* x32, const float* F, const float* c_skip,
                const float* c_out, const double* sigma_div, double* eps, double* denoised,
                int B, int64_t D, void* stream);
/* out[b,:] = ca[b]*x[b,:] (+ cb[b]*y[b,:])   the Euler / Heun updates, eps mixing and scaling (:884-917) */
int nlc_f64_lincomb(const double* x, const double* ca, const double* y, const double* cb,
                    double* out, int B, int64_t D, void* stream);

/* ------------------------------------------------------------------------------------
 * Inpainting measurement operator (functions/svd_operators.py:324-359, Inpainting.A / A_pinv through
 * V/Vt/U/Ut/add_zeros at :52-58,68-80) on the NCHW f32 state.  Indices live in the reference's
 * pixel-interleaved flattening  idx = p*C + c.
 *   nlc_inpaint_A     : out[b][j] = x[b][c][p]                    kept[j] = p*C + c   (int64, sorted)
 *   nlc_inpaint_Apinv : out[b][c][p] = inv[p*C+c] >= 0 ? y[b][inv[p*C+c]] : 0        (inv: int32 [HW*C])
 * The per-step projection x0 <- x0 - A^+(A x0 - y) is the mask/known path of nlc_sched_step.
 * ---------------------------------------------------------------------------------- */
int nlc_inpaint_A(const float* x, const int64_t* kept, float* out, int B, int C, int64_t HW,
                  int64_t nk, void* stream);
int nlc_inpaint_Apinv(const float* y, const int32_t* inv, float* out, int B, int C, int64_t HW,
                      int64_t nk, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NLC_HIP_H */
