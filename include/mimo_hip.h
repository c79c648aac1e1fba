/*
 * mimo_hip.h — C-ABI of libmimo_hip.so: the MI355X (gfx950) kernels behind the MIMO
 * denoising hot path (reference_unet + denoising_unet + sd-vae-ft-mse + pose guider).
 *
 * The reference (menyifang/MIMO) has NO native code and NO FFI on this path: every op
 * is a stock torch/diffusers call made from Python.  The entry points
 * below are therefore what a maintainer would bind (ctypes) *in
 * place of* those stock ops; each one cites the reference call site it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers + sizes; no torch types.  All pointers are DEVICE pointers that
 *     the caller allocated (torch.empty(...).data_ptr()); the library never allocates,
 *     frees or retains device memory, reads no environment variable and keeps no mutable
 *     global state (every launch is a pure function of its arguments).
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - return 0 on success; negative MIMO_E* for a rejected argument; positive values
 *     are hipError_t from the launch.
 *   - activations are channels-last "token-major": an image batch [n, H, W, C] is the
 *     row-major matrix [n*H*W, C].  `half16` tensors hold IEEE fp16 or bf16 according
 *     to the `dtype` argument (MIMO_F16 / MIMO_BF16); MFMA accumulation is fp32.
 */
#ifndef MIMO_HIP_H
#define MIMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIMO_OK 0
#define MIMO_EINVAL (-1)   /* bad shape / alignment / flag combination */
#define MIMO_EDTYPE (-2)   /* unknown dtype code */
#define MIMO_EARCH (-3)    /* no gfx950 device */

#define MIMO_F16 0
#define MIMO_BF16 1
#define MIMO_F32 2   /* accepted by mimo_u8_to_tokens only (an fp32 image for the VAE's split-operand policy) */

/* epilogue flags shared by mimo_gemm / mimo_conv2d */
#define MIMO_EPI_SILU 1u      /* v = silu(v) after the bias terms, before the residual */
#define MIMO_EPI_GEGLU 2u     /* weight rows interleaved [16 value | 16 gate]; out = value*gelu_erf(gate), N_out = N/2 */
#define MIMO_EPI_OUT_F32 4u   /* store fp32 (else half16) */
#define MIMO_EPI_RES_F32 8u   /* residual tensor is fp32 (else half16) */
#define MIMO_EPI_NO_SPLITK 16u /* never split the K reduction of THIS call (its fp32 summation order then does not
                                  depend on the row count M, i.e. on how a clip is cut into batches) */

int mimo_version(void);

/* Workspace convention (mimo_gemm / mimo_conv2d): the caller MAY pass a 16-byte-aligned device scratch buffer.  It is
 * only used to split a long K reduction over several workgroups when the output has too few tiles to fill the chip
 * (fp32 partial tiles, reduced in a fixed order by a second launch on the same stream); with workspace == NULL or
 * workspace_bytes too small the call runs unsplit.  mimo_workspace_bytes() is sufficient for every call with
 * M * N <= 2^22 output elements (the SD1.5 8x8 level at 48 images); the library never allocates device memory. */
size_t mimo_workspace_bytes(void);

/* No-op kept for tools/microbench.py.  The shipped library reads no environment variable and has no mutable global
 * state: every tuning knob is a compile-time constant.  (The separate -DMIMO_TUNE build of the same sources,
 * libmimo_hip_tune.so, reads MIMO_* knobs from the environment at every launch for interleaved A/B timing.)  Returns 0. */
int mimo_reload_tuning(void);

/* ---------------------------------------------------------------------------------
 * mimo_gemm: out[M,N] = epi( A[M,K] . W[N,K]^T )
 *   replaces every nn.Linear / 1x1 nn.Conv2d on the path:
 *     attention to_q/to_k/to_v/to_out      (diffusers Attention; called src/models/attention.py:321-345,
 *                                            src/models/mutual_self_attention.py:154-197)
 *     GEGLU feed-forward                    (diffusers FeedForward; src/models/attention.py:429,
 *                                            src/models/motion_module.py:258)
 *     Transformer3DModel proj_in/proj_out   (src/models/transformer_3d.py:124-130,148-165)
 *     motion-module proj_in/proj_out        (src/models/motion_module.py:158,172)
 *     time embedding + time_emb_proj        (src/models/unet_3d_edit_bkfill.py:462-468, src/models/resnet.py:226)
 *   A: half16 row-major, leading dimension lda (elements).  W: half16 [N,K] row-major
 *   (torch Linear.weight layout).  K % 8 == 0, lda % 8 == 0, 16-byte aligned pointers.
 *   epilogue: v = acc; +bias[n] (fp32, nullable); +img_bias[(m / rows_per_img) * img_bias_ld + n]
 *   (fp32, nullable; one row per group of rows_per_img consecutive output rows: the
 *   time-embedding projection / collapsed 1-key cross-attention of a batch element);
 *   SILU; +residual[m, ldr] (nullable); *out_scale; store to out[m, ldo].
 * --------------------------------------------------------------------------------- */
int mimo_gemm(int dtype, const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
              int64_t M, int N, int K, const float* bias, const float* img_bias,
              int64_t img_bias_ld, int64_t rows_per_img, const void* residual, int64_t ldr,
              float out_scale, unsigned flags, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused side outputs of the GEMM / convolution epilogue (mimo_gemm_ext / mimo_conv2d_ext; ext == NULL is the
 * plain call).  They remove whole HBM passes of the normalisation layers that FOLLOW the producing op:
 *   colstats  fp32 [M/32][2][N] (M % 32 == 0): for every 32-row slab and column, the mean and the sum of squared
 *             deviations from that mean of the values this call stores.  mimo_group_norm_stats_cols() merges them
 *             into GroupNorm (mean, rstd) — the statistics pass of InflatedGroupNorm / nn.GroupNorm
 *             (src/models/resnet.py:20-28, transformer_3d.py:124, motion_module.py:154) never re-reads the tensor.
 *             A call with colstats never splits K.
 *   ln_out    half16 [M, N]: LayerNorm(out row) * ln_gamma + ln_beta (+ ln_pe[(m / ln_rows_per_frame) % ln_pe_frames])
 *             of every output row, eps ln_eps — the nn.LayerNorm (+ PositionalEncoding) that consumes this GEMM's
 *             result (src/models/attention.py:329-360,391-429; src/models/motion_module.py:230-258,276-279).
 *             mimo_gemm_ext only; needs N == 320 (one tile holds whole rows), ldo == N, no SILU/GEGLU, and with
 *             ln_pe: ln_rows_per_frame % 128 == 0.  Other shapes: call mimo_layer_norm.
 *   LayerNorm folded into the CONSUMING projection (C = 640 / 1280, where no tile holds a whole row; the same
 *   nn.LayerNorm sites as ln_out): LayerNorm(x) . W^T + b = rstd * (x . (gamma o W)^T - mean * colsum) + (beta . W^T + b).
 *     producer (the GEMM that writes x: proj_in / to_out + residual; fp32 out, no SILU / GEGLU, mimo_row_stat_slots(N) > 0):
 *       row_half   half16 [M, N]: the stored rows rounded to the operand type — the consumer's A operand
 *       row_stats  fp32 [M][mimo_row_stat_slots(N)][2]: per row and column slot (sum, sum of squares) of the rounded values
 *     consumer (to_q/k/v, GEGLU proj; any epilogue): A = row_half, W = half(gamma o W), bias = beta . W^T + b and
 *       a_row_stats (the producer's row_stats), a_slots = mimo_row_stat_slots(K), a_colsum fp32 [N] = sum_k W[n, k] of
 *       the ROUNDED weight, a_eps: v = rstd[m] * (acc - mean[m] * a_colsum[n]) in front of the ordinary epilogue.
 *   Neither half splits K.  The slot layout is a function of N alone, so a row's statistics do not depend on the batch.
 * --------------------------------------------------------------------------------- */
typedef struct mimo_epilogue_ext {
  float* colstats;
  void* ln_out;
  const float* ln_gamma;
  const float* ln_beta;
  const float* ln_pe;
  float ln_eps;
  int ln_pe_frames;
  int64_t ln_rows_per_frame;
  void* row_half;
  float* row_stats;
  const float* a_row_stats;
  const float* a_colsum;
  int a_slots;
  float a_eps;
} mimo_epilogue_ext;

/* number of column slots of row_stats for a producer of width N (0: the folded form does not cover N).  Covered: N a whole
 * number of the producer family's widest tile (N % 256 == 0 with 64-column slots, N % 320 == 0 with 80-column slots: the row-side
 * epilogue has no column mask) and at most 20 slots (what a consumer row holds) — 640 -> 8, 1280 -> 20; 960, 1920, 2560 -> 0.
 * A consumer's a_slots must equal mimo_row_stat_slots(K); anything else is MIMO_EINVAL. */
int mimo_row_stat_slots(int N);

int mimo_gemm_ext(int dtype, const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                  int64_t M, int N, int K, const float* bias, const float* img_bias,
                  int64_t img_bias_ld, int64_t rows_per_img, const void* residual, int64_t ldr,
                  float out_scale, unsigned flags, void* workspace, size_t workspace_bytes,
                  const mimo_epilogue_ext* ext, void* stream);

/* ---------------------------------------------------------------------------------
 * mimo_conv2d: channels-last implicit-GEMM convolution (3x3 or 1x1), MFMA.
 *   replaces InflatedConv3d / nn.Conv2d + the fused adds around it:
 *     ResnetBlock3D conv1 (+ time_emb_proj add), conv2 (+ shortcut add)   src/models/resnet.py:217-247
 *     Downsample3D (stride 2), Upsample3D (nearest + conv)                src/models/resnet.py:53-120
 *     conv_in (+ pose_cond_fea add), conv_out                             src/models/unet_3d_edit_bkfill.py:483-485,569-571
 *     PoseGuider convs (+ SiLU)                                           src/models/pose_guider.py:47-57
 *     diffusers ResnetBlock2D/Downsample2D/Upsample2D and the VAE Encoder/Decoder convs
 *   in:  half16 [n, Hin, Win, Cin]; Cin % 8 == 0.
 *   W:   half16 [Cout, ks*ks*Cin (+ Cin2)], K index = (ky*ks + kx)*Cin + ci, then the
 *        optional extra 1x1 tap over in2.
 *   in2: optional second source half16 [n, Hout, Wout, Cin2] contributing an extra
 *        1x1 tap (the ResBlock conv_shortcut fused as additional K).
 *   Spatial map: output (oy,ox) reads virtual input (oy*stride - pad_t + ky, ...);
 *   the virtual input is `in` itself, or its nearest-neighbour upsampling to
 *   [Hup, Wup] when Hup > 0 (src index = floor(dst * scale), scale = Hin/Hup as
 *   float, exactly torch's 'nearest').  Zero padding outside.
 *   epilogue: as mimo_gemm with rows_per_img = Hout*Wout*imgs_per_bias_row.
 * --------------------------------------------------------------------------------- */
typedef struct mimo_conv_params {
  int n, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int ksize;          /* 1 or 3 */
  int stride;         /* 1 or 2 */
  int pad_t, pad_l;   /* zero padding before the first row / column */
  int Hup, Wup;       /* 0,0 = no upsampling; else virtual input size */
  int Cin2;           /* channels of in2 (0 = none) */
  int imgs_per_bias_row; /* img_bias row = image / imgs_per_bias_row (frames of one batch element); 0 -> 1 */
  int img_bias_ld;       /* row pitch of img_bias in floats; 0 -> Cout */
} mimo_conv_params;

int mimo_conv2d(int dtype, const void* in, const void* in2, const void* W, void* out,
                const mimo_conv_params* p, const float* bias, const float* img_bias,
                const void* residual, float out_scale, unsigned flags, void* workspace,
                size_t workspace_bytes, void* stream);
/* Thin-output 3x3 convolution (Cout <= 16: UNet conv_out, src/models/unet_3d_edit_bkfill.py:560-563; VAE decoder conv_out,
 * diffusers AutoencoderKL.decode via pipeline_..._roiclip.py:120) as GEMM + gather: taps = mimo_gemm(in[M, Cin],
 * Wt[9 Cout, Cin]) (Wt row tap * Cout + c = weight[c, :, ky, kx]: mimo_amd.packing.pack_conv_taps; fp32 out) holds every
 * pixel's contribution to the nine outputs around it, the input is read once; this call sums them:
 *   out[n, y, x, c] (fp32) = (bias[c] + sum_{ky,kx} taps[(n, y + ky - 1, x + kx - 1)][(3 ky + kx) Cout + c]) * out_scale
 * (padding = 1, stride 1, fixed tap order).  taps: fp32 [n*H*W, ldt], ldt >= 9 Cout; Cout in {4, 8, 16}. */
int mimo_conv3x3_tapsum(const float* taps, int64_t ldt, int n, int H, int W, int cout, const float* bias, float* out,
                        float out_scale, void* stream);
/* as mimo_conv2d, plus ext->colstats (ext->ln_out must be NULL) */
int mimo_conv2d_ext(int dtype, const void* in, const void* in2, const void* W, void* out,
                    const mimo_conv_params* p, const float* bias, const float* img_bias,
                    const void* residual, float out_scale, unsigned flags, void* workspace,
                    size_t workspace_bytes, const mimo_epilogue_ext* ext, void* stream);

/* ---------------------------------------------------------------------------------
 * GroupNorm (per image, 32 groups typical) over a *virtual channel concat* of two
 * sources (torch.cat([h, skip], dim=1) is never materialised, src/models/unet_3d_blocks.py:697,827).
 *   replaces InflatedGroupNorm / nn.GroupNorm (+ F.silu)  src/models/resnet.py:20-28,220-221,231,237;
 *   src/models/transformer_3d.py:124; src/models/motion_module.py:154; VAE norms.
 *   x1: [n, HW, C1], x2: [n, HW, C2] (nullable, C2 = 0); each fp32 or half16 (x_is_f32).
 *   stats: fp32 [n, groups, 2] = (mean, rstd).  split >= 1 pixel slices per (image, group) with
 *   partials: caller scratch fp32 [n*groups*split, 2] (split > 1 only; reduced in fixed order).  apply: y = (x-mean)*rstd*gamma+beta,
 *   optional SiLU, stored half16 [n, HW, C1+C2].  raw_out (nullable): plain half16 cast
 *   of the concatenated input (feeds the fused 1x1 shortcut / upsample conv).
 * --------------------------------------------------------------------------------- */
int mimo_group_norm_stats(const void* x1, int C1, const void* x2, int C2, int x_is_f32, int dtype,
                          int n, int64_t HW, int groups, float eps, float* stats, float* partials,
                          int split, void* stream);
/* GroupNorm statistics from epilogue column statistics (mimo_epilogue_ext.colstats) of the one or two tensors of the
 * virtual concat: cs1 fp32 [n*HW/32][2][C1], cs2 [n*HW/32][2][C2] (nullable, C2 = 0); HW % 32 == 0.  One wave per
 * (image, group) merges slabs x columns with Chan's parallel variance update in double, fixed order.
 * stats: fp32 [n, groups, 2] = (mean, rstd), exactly what mimo_group_norm_apply consumes. */
int mimo_group_norm_stats_cols(const float* cs1, int C1, const float* cs2, int C2, int n, int64_t HW,
                               int groups, float eps, float* stats, void* stream);
/* The same merge for partials over `rows_per_slab` pixels each (HW % rows_per_slab == 0): 32 = the column statistics of
 * mimo_gemm_ext / mimo_conv2d_ext / the fused tails / mimo_conv3x3_fused with a residual, 256 = the tile statistics of
 * mimo_conv3x3_fused without one.  The two sources of a virtual concat may use different slab sizes (rows_per_slab2 is
 * ignored when C2 = 0); slabs are image-major, any pixel order inside an image. */
int mimo_group_norm_stats_slabs(const float* cs1, int C1, int rows_per_slab1, const float* cs2, int C2, int rows_per_slab2,
                                int n, int64_t HW, int groups, float eps, float* stats, void* stream);
int mimo_group_norm_apply(const void* x1, int C1, const void* x2, int C2, int x_is_f32, int dtype,
                          int n, int64_t HW, int groups, const float* stats, const float* gamma,
                          const float* beta, int silu, void* out, void* raw_out, void* stream);
/* Split-operand apply pass (the VAE's "split" precision policy; diffusers AutoencoderKL.encode as called at
 * src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:427-439, whose fp32 result the 1e-3 parity bar is measured
 * against): y = GroupNorm(x) (+ SiLU) — or y = x when stats == NULL — of an fp32 tensor x [n, HW, C] is stored as THREE
 * half16 channel blocks per token, out[tok][0..C) = out[tok][C..2C) = hi = half(y), out[tok][2C..3C) = lo = half(y - hi);
 * row stride ldo >= 3 C elements (columns beyond 3 C are left untouched: the caller zero-fills padding once).  Multiplied by a
 * weight packed [Whi | Wlo | Whi] along K by the ordinary mimo_gemm / mimo_conv2d the fp32 accumulators receive
 * hi.Whi + hi.Wlo + lo.Whi: both operands carry ~22 mantissa bits.  C % 8 == 0, ldo % 8 == 0, 16-byte aligned pointers.
 * x may be one source of a virtual channel concat (the skip connections, src/models/unet_3d_blocks.py:697,827): c_total > 0 =
 * channels of the concatenated tensor that `groups`, `stats`, `gamma`, `beta` describe, c_off = this source's first channel
 * in it (c_total <= 0: x is the whole tensor). */
int mimo_group_norm_apply_split3(const float* x, int C, int dtype, int n, int64_t HW, int groups, const float* stats,
                                 const float* gamma, const float* beta, int silu, void* out, int64_t ldo, int c_off, int c_total,
                                 void* stream);
/* GroupNorm folded to a per-(image, channel) affine: ab fp32 [n][2][C], ab[i][0][c] = rstd(i, g(c)) * gamma[c],
 * ab[i][1][c] = beta[c] - mean(i, g(c)) * ab[i][0][c], so that GroupNorm(x)[c] = x * a + b.  The operand of
 * mimo_conv3x3_fused (the apply pass of src/models/resnet.py:20-28,220-221,237 without a pass over the tensor). */
int mimo_group_norm_affine(const float* stats, const float* gamma, const float* beta, int n, int C, int groups,
                           float* ab, void* stream);

/* ---------------------------------------------------------------------------------
 * mimo_conv3x3_fused: 3x3 / stride 1 / pad 1 convolution that normalises its own input.
 *   out = epilogue( conv3x3( silu?( x * a + b ) ) ),   x = the fp32 virtual channel concat [x1 | x2]
 *   replaces, in ONE launch, the GroupNorm-apply (+ SiLU) pass AND the convolution it feeds:
 *     ResnetBlock3D norm1 -> nonlinearity -> conv1 (+ time_emb_proj add)          src/models/resnet.py:217-229
 *     ResnetBlock3D norm2 -> nonlinearity -> conv2 (+ residual, / output_scale)   src/models/resnet.py:231-247
 *     diffusers ResnetBlock2D of the VAE Encoder / Decoder (AutoencoderKL, pipeline_...roiclip.py:120,430,438)
 *     Upsample3D / Upsample2D: nearest x2 -> conv (upsample2x = 1, ab = NULL: plain cast)   src/models/resnet.py:31-76
 *   x1: fp32 [n, Hs, Ws, C1], x2: fp32 [n, Hs, Ws, C2] or NULL (C2 = 0); (Hs, Ws) = (H, W), or (H/2, W/2) with upsample2x.
 *       (C1 + C2) % 32 == 0; with x2: C1 % 64 == 0.  Only fp32 inputs: the residual stream is fp32.
 *   ab: fp32 [n][2][C] from mimo_group_norm_affine, or NULL (no normalisation); silu: apply SiLU after the affine.
 *   W:  half16 [Cout][ldw], K index = (ky*3 + kx) * C + c (the mimo_conv2d packing; ldw >= 9 C, extra columns —
 *       a fused shortcut segment — are ignored).
 *   out: fp32 [n, H, W, Cout]; H % 16 == 0, W % 16 == 0 (a block owns a 16 x 16 pixel tile + halo).
 *   raw_out (nullable, not with upsample2x): half16 [n, H, W, C] plain cast of x, written as a side effect (it feeds
 *       the fused 1x1 shortcut of the block's second convolution through mimo_conv2d's in2).
 *   tile_stats (nullable): fp32 [n * (H/16) * (W/16)][2][Cout]: per 16 x 16 pixel tile and output channel, the mean and the
 *       sum of squared deviations from that mean of the values this call stores (exact two-pass inside the tile, fixed
 *       order).  mimo_group_norm_stats_cols(..., rows_per_slab = 256) merges them into GroupNorm (mean, rstd): the
 *       norm that consumes `out` makes no statistics pass over HBM.  Not with a residual / MIMO_EPI_SILU (MIMO_EINVAL).
 *   epilogue: bias [Cout], img_bias row = image / imgs_per_bias_row (the time embedding), MIMO_EPI_SILU,
 *       residual fp32 [n, H, W, Cout], out_scale.  flags must contain MIMO_EPI_OUT_F32 (and MIMO_EPI_RES_F32 with a
 *       residual).  Returns MIMO_EINVAL for shapes it does not cover (the caller then takes the two-launch path);
 *       with ab: C <= 960 when Cout % 320 == 0, C <= 2560 otherwise (the affine table lives in LDS).
 *   The summation order of an output element depends on the layer only, never on n.
 * --------------------------------------------------------------------------------- */
typedef struct mimo_hconv_params {
  int n, H, W, Cout;
  int upsample2x;        /* 0 | 1 */
  int imgs_per_bias_row; /* 0 -> 1 */
  int img_bias_ld;       /* row pitch of img_bias in floats; 0 -> Cout */
} mimo_hconv_params;

int mimo_conv3x3_fused(int dtype, const float* x1, int C1, const float* x2, int C2, const float* ab, int silu,
                       const void* W, int64_t ldw, float* out, const mimo_hconv_params* p, const float* bias,
                       const float* img_bias, const float* residual, void* raw_out, float* tile_stats,
                       float out_scale, unsigned flags, void* stream);

/* ---------------------------------------------------------------------------------
 * LayerNorm over the last dim of x [rows, C] (fp32 or half16) -> half16 `out` and / or fp32 `out_f32`, optional
 * additive positional table pe[(row / rows_per_frame) % pe_frames, C] (fp32).
 *   replaces nn.LayerNorm (src/models/attention.py:329-360; src/models/motion_module.py:230-258)
 *   and PositionalEncoding.forward (src/models/motion_module.py:276-279).
 * --------------------------------------------------------------------------------- */
int mimo_layer_norm(const void* x, int x_is_f32, int dtype, int64_t rows, int C, float eps,
                    const float* gamma, const float* beta, const float* pe, int64_t rows_per_frame,
                    int pe_frames, void* out, float* out_f32, void* stream);

/* ---------------------------------------------------------------------------------
 * The GEGLU feed-forward of a transformer block in ONE launch (C = 320, the full-resolution level of the UNets):
 *   out[M, C] (half16) = residual[M, C] (fp32) + GEGLU(A[M, C] @ W1^T + b1) @ W2^T + b2
 *   replaces diffusers FeedForward([GEGLU(dim, 4 dim), Dropout, Linear(4 dim, dim)]) + the residual add of
 *   src/models/attention.py:428-429 (spatial blocks, via mutual_self_attention.py:232-239) and
 *   src/models/motion_module.py:258 (temporal blocks): the [M, 4C] GEGLU intermediate stays on the chip.
 *   A: half16 LayerNorm output; W1: half16 [8C, C] GEGLU-packed as for mimo_gemm (16 value rows | 16 gate rows blocks),
 *   b1 packed alike; W2: half16 [C, 4C] with its K axis permuted inside every 32-block (mimo_amd.packing.pack_ff2_kperm:
 *   position 8g + j <- 4g + j | 16 + 4g + (j - 4)); b2 fp32 [C].  MIMO_EINVAL unless C == 320.
 * --------------------------------------------------------------------------------- */
int mimo_ff_fused(int dtype, const void* A, int64_t lda, const void* W1, const float* b1, const void* W2,
                  const float* b2, const float* residual, int64_t ldr, void* out, int64_t ldo, int64_t M, int C,
                  void* stream);
/* The same with the block's output projection and its residual folded in (the end of Transformer3DModel.forward,
 * src/models/transformer_3d.py:150-169, and of TemporalTransformer3DModel.forward, src/models/motion_module.py:170-184):
 *   out[M, C] (fp32) = x[M, C] (fp32, the block input) + (residual + FF(A)) @ Wp^T + bp
 *   Wp: half16 [C, C] = proj_out.weight with rows in tile order and the K axis permuted (mimo_amd.packing.pack_proj_tail).
 *   colstats: NULL, or fp32 [M / 32, 2, C] (M % 32 == 0): the launch also writes the GroupNorm column statistics of `out`
 *   per 32-row slab — (mean, sum of squared deviations) per column, the layout of mimo_epilogue_ext.colstats — from the
 *   values in registers, so the GroupNorm that consumes `out` (src/models/motion_module.py:156, src/models/resnet.py:217)
 *   makes no statistics pass; merge with mimo_group_norm_stats_slabs(rows_per_slab = 32). */
int mimo_ff_proj_fused(int dtype, const void* A, int64_t lda, const void* W1, const float* b1, const void* W2,
                       const float* b2, const float* residual, int64_t ldr, const void* Wp, const float* bp,
                       const float* x, int64_t ldx, float* out, int64_t ldo, int64_t M, int C, float* colstats,
                       void* stream);
/* The whole tail of a transformer block after its attention core in one launch: the attention output projection with its
 * residual (+ the collapsed cross-attention as a per-image vector), the LayerNorm in front of the feed-forward, the
 * feed-forward and the owning transformer's proj_out with its residual (BasicTransformerBlock.forward after attn1,
 * src/models/attention.py:208-262, + Transformer3DModel.forward's end, src/models/transformer_3d.py:150-169; the
 * TemporalTransformerBlock after its second attention, src/models/motion_module.py:240-262 + :170-184):
 *   y = residual + O @ Wo^T + bo (+ img_bias[row / rows_per_img]);  n = LayerNorm(y) * ln_gamma + ln_beta
 *   out[M, C] (fp32) = x + (y + FF(n)) @ Wp^T + bp
 *   O: half16 attention output; Wstream: half16 [10 C, C] = [Wo with rows in tile order (pack_rows_tail) | W1 GEGLU-packed
 *   with its K axis permuted (pack_ff2_kperm) | Wp as for mimo_ff_proj_fused] (mimo_amd.packing.pack_block_tail_stream);
 *   y and n never reach memory.  img_bias: fp32 [ceil(M / rows_per_img), ldib] or NULL, rows_per_img >= 128.
 *   colstats: as for mimo_ff_proj_fused.  MIMO_EINVAL unless C == 320.
 *   Kernel: ff4_kernel (csrc/ff_tail4.hip, four waves x 512 registers); it evaluates the two residual sums as (o Wo^T + bo + img_bias)
 *   + residual and (z Wp^T + bp) + x — the same numbers in a different fp32 order than written above. */
int mimo_block_tail_fused(int dtype, const void* O, int64_t ldo_in, const void* Wstream, const float* bo,
                          const float* img_bias, int64_t ldib, int64_t rows_per_img, const float* residual, int64_t ldr,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, const float* b1, const void* W2,
                          const float* b2, const float* bp, const float* x, int64_t ldx, float* out, int64_t ldo,
                          int64_t M, int C, float* colstats, void* stream);
/* Everything a transformer block does BEFORE its attention core in one launch (round 5): the projection that produces the
 * token stream, the LayerNorm in front of the attention and the fused Q/K/V projection —
 *   spatial block:   GroupNorm -> proj_in (Transformer3DModel.forward, src/models/transformer_3d.py:118-140) -> norm1
 *                    (TemporalBasicTransformerBlock.forward, src/models/attention.py:329-360 /
 *                    mutual_self_attention.py:118-120) -> attn1.to_q / to_k / to_v
 *   motion module:   GroupNorm -> proj_in (TemporalTransformer3DModel.forward, src/models/motion_module.py:150-163) -> norms[0]
 *                    + positional encoding -> attention_blocks[0].to_q/k/v (:230-248, :300-330); and for the second attention
 *                    attention_blocks[0].to_out + residual -> norms[1] + PE -> attention_blocks[1].to_q/k/v
 *   y[M, C] (fp32)  = (residual +) A' @ Wi^T + bi                      (written: the residual of the attention's to_out)
 *   qkv[M, 3C] (half) = (LayerNorm(y) * ln_gamma + ln_beta (+ ln_pe[(m / ln_rows_per_frame) % ln_pe_frames])) @ Wqkv^T
 *   A' = A (half16 [M, lda]: an attention output), or — A == NULL — half(x32 * a + b) with x32 the fp32 block input and
 *   gn_ab = fp32 [M / rows_per_img, 2, C] the GroupNorm folded to a per-(image, channel) affine (mimo_group_norm_affine):
 *   the normalised tensor never reaches memory (rows_per_img >= 128: a 128-row panel holds rows of at most two images; the
 *   reference's default 784 x 784 frames give 9604 rows per image).
 *   Wstream: half16 [4C, C] = [Wi with rows in tile order (pack_rows_tail) | [Wq; Wk; Wv] with its K axis permuted
 *   (pack_ff2_kperm)] (mimo_amd.packing.pack_block_head_stream); the LayerNorm output never reaches memory.
 *   ln_pe: NULL or fp32 [ln_pe_frames, C], ln_rows_per_frame >= 128.  MIMO_EINVAL unless C == 320. */
int mimo_block_head_fused(int dtype, const void* A, int64_t lda, const float* x32, int64_t ldx, const float* gn_ab,
                          int64_t rows_per_img, const void* Wstream, const float* bi, const float* residual, int64_t ldr,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, const float* ln_pe,
                          int64_t ln_rows_per_frame, int ln_pe_frames, float* y_out, int64_t ldy, void* qkv, int64_t ldq,
                          int64_t M, int C, void* stream);

/* ---------------------------------------------------------------------------------
 * Spatial multi-head attention (flash, online softmax, MFMA 32x32x16) with an optional
 * second key/value segment shared by all batch rows b >= seg2_first_batch
 * (the reference-attention bank):
 *   replaces diffusers Attention/AttnProcessor2_0 SDPA as called in read mode
 *   (src/models/mutual_self_attention.py:154-197: cond rows attend [self || bank], uncond
 *   rows attend self) and in write mode / plain self-attention (:137-147).
 *   q,k,v: half16 token-major [B, Nq|Nk, ld*] with head h at columns [h*d, (h+1)*d);
 *   k2,v2: half16 [Nk2, ld*2] (nullable).  out: half16 [B, Nq, ldo].  d in {40, 64, 80, 160}, or d = 512 with
 *   heads = 1 and no second segment: the single-head mid-block attention of the VAE (diffusers UNetMidBlock2D
 *   Attention over the H*W latent tokens; AutoencoderKL, pipeline_pose2vid_long_edit_bkfill_roiclip.py:120,430).
 *   scale > 0: softmax(scale * q.k^T).  scale <= 0: q is ALREADY multiplied by scale * log2(e) (the
 *   caller folded it into W_q); for d = 40 this selects the kernel variant whose softmax has no
 *   per-score multiply/subtract (the running max rides in a spare MFMA k-slot).
 * --------------------------------------------------------------------------------- */
int mimo_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                   int64_t ldv, const void* k2, int64_t ldk2, const void* v2, int64_t ldv2, void* out,
                   int64_t ldo, int B, int Nq, int Nk, int Nk2, int seg2_first_batch, int heads, int d,
                   float scale, void* stream);
/* The same operation with Q.K^T on the fp8 (e4m3, OCP) MFMA — BASELINE configs[4] "fp8 MFMA attention QK", opt-in,
 * accuracy reported, not gated.  Q and K are converted to fp8 inside the kernel (K once per tile while
 * it is staged, Q once per block); softmax and P.V stay on half operands.  d in {40, 80, 160}.  Same arguments as
 * mimo_attention (same reference op: src/models/mutual_self_attention.py:154-197). */
int mimo_attention_fp8qk(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v,
                         int64_t ldv, const void* k2, int64_t ldk2, const void* v2, int64_t ldv2, void* out,
                         int64_t ldo, int B, int Nq, int Nk, int Nk2, int seg2_first_batch, int heads, int d,
                         float scale, void* stream);

/* ---------------------------------------------------------------------------------
 * Temporal attention over the frame axis without materialising '(b f) d c -> (b d) f c'
 *   replaces VersatileAttention.forward (src/models/motion_module.py:353-390).
 *   q,k,v: half16 [b*F, HW, ld*] (frame-major tokens); attention is over the F frames
 *   of each (batch b, pixel p, head h).  F <= 32.  out: half16 [b*F, HW, ldo].
 * --------------------------------------------------------------------------------- */
int mimo_temporal_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                            const void* v, int64_t ldv, void* out, int64_t ldo, int b, int F,
                            int64_t HW, int heads, int d, float scale, void* stream);

/* row softmax: in fp32 [rows, cols] * scale -> half16 [rows, ldo] (VAE mid-block attention,
 * 1 head d=512: scores via mimo_gemm, softmax here, P.V via mimo_gemm). */
int mimo_softmax_rows(int dtype, const float* in, int64_t ldi, void* out, int64_t ldo, int64_t rows,
                      int cols, float scale, void* stream);

/* ---------------------------------------------------------------------------------
 * layout / elementwise helpers at the boundary
 * --------------------------------------------------------------------------------- */
/* [b, C, F, H, W] (fp32 or half16 per in_is_f32) frame-gathered -> half16 [b*F', H*W, Cpad] with
 * channels >= C zero; frame_idx (int32 [F'] device, nullable = identity).  Replaces the
 * `rearrange "b c f h w -> (b f) c h w"` churn (src/models/resnet.py:13-15) at the model input. */
int mimo_ncfhw_to_tokens(const void* in, int in_is_f32, int dtype, int b, int C, int F, int H, int W,
                         const int* frame_idx, int Fsel, int Cpad, int64_t out_ld, int out_col0,
                         void* out, void* stream);
/* tokens (fp32 or half16) [b*F, H*W, ld] cols [0,C) -> fp32 [b, C, F, H, W] */
int mimo_tokens_to_ncfhw(const void* in, int in_is_f32, int dtype, int64_t ld, int b, int C, int F,
                         int H, int W, float scale, float* out, void* stream);
/* half16/fp32 cast with optional second output */
int mimo_cast(const void* in, int in_is_f32, int dtype, int64_t count, void* out_half, void* stream);

/* ---------------------------------------------------------------------------------
 * Window accumulate + classifier-free guidance + DDIM (v-prediction, eta = 0) step
 *   replaces src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:540-553
 *   (noise_pred/counter, chunk(2), guidance, scheduler.step) in one pass.
 *   acc: fp32 [2, 4, F, h, w] summed window predictions (row 0 uncond, row 1 cond; if
 *   !cfg: [1,...]).  counter: fp32 [F].  latents: fp32 [1,4,F,h,w], updated in place.
 * --------------------------------------------------------------------------------- */
int mimo_cfg_ddim_step(const float* acc, const float* counter, float* latents, int C, int F,
                       int64_t HW, int cfg, float guidance, float sqrt_a_t, float sqrt_1ma_t,
                       float sqrt_a_prev, float sqrt_1ma_prev, void* stream);
/* The same update for the listed frames only (frames: int32 [nf] device, distinct, each in [0, F)); per element the
 * arithmetic of mimo_cfg_ddim_step bit for bit.  The sharded long clip's cross-step schedule advances a frame from step t
 * to t + 1 as soon as every context window that covers it (src/pipelines/context.py:15-42) has delivered its step-t
 * prediction — the reference's loop (:505-553) has no dependency between frames that share no window. */
int mimo_cfg_ddim_step_frames(const float* acc, const float* counter, float* latents, int C, int F, int64_t HW,
                              const int* frames, int nf, int cfg, float guidance, float sqrt_a_t, float sqrt_1ma_t,
                              float sqrt_a_prev, float sqrt_1ma_prev, void* stream);
/* acc[:, :, frames[j]] += pred[:, :, j]; counter[frames[j]] += 1, with pred given as the UNet's
 * token-major output fp32 [bb*Fw, HW, ld] (channels [0,C)); frames: int32 [Fw] device; frames[j] < 0: row j of the
 * prediction is skipped (only a frame segment of the window is taken). */
int mimo_window_accumulate(const float* pred, int64_t ld, const int* frames, int Fw, int bb, int C,
                           int F, int64_t HW, float* acc, float* counter, void* stream);

/* differ[i] = 1 where frame i of `frames` ([n] frames of frame_bytes bytes each, any element type, frame_bytes % 16 == 0)
 * is not bit-identical to frame i - 1; differ[0] = 1.  The caller zero-fills differ[1..n) first (int32 [n] device).
 * run_animate.py feeds the VAE encoder F copies of ONE background frame (tools/util.py:339-345 `init_bk`, encoded one by
 * one at pipeline :436-443): the encoder runs once per run of identical frames. */
int mimo_frames_differ(const void* frames, int n, int64_t frame_bytes, int* differ, void* stream);

/* VAE image post-process: tokens half16/fp32 [n, H*W, ld] (3 ch) -> fp32 [n,3,H,W] = clamp(x/2+0.5,0,1)
 *   (src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:123) */
int mimo_tokens_to_image(const void* in, int in_is_f32, int dtype, int64_t ld, int n, int H, int W,
                         float* out, void* stream);


/* ---------------------------------------------------------------------------------
 * Image kernels either side of the denoising path: byte-exact integer / IEEE work.
 * --------------------------------------------------------------------------------- */
/* One separable pass of Pillow's 8-bit resampler (PIL.Image.resize; Pillow src/libImaging/Resample.c):
 *   dst[img, y, x, c] = clip8((2^21 + sum_j src[.. first + j ..] * coeffs[o][j]) >> 22), o = x (horizontal) or y.
 * bounds int32 [out][2] = (first source index, tap count), coeffs int32 [out][ksize]: Pillow's precompute_coeffs +
 * normalize_coeffs_8bpc, computed on the host (mimo_amd/image.py).  A horizontal then a vertical pass equal
 * Image.resize bit for bit.  src: uint8, or fp32 quantised on the fly as (uint8)(v * 255.0f) — the
 * `(image * 255).astype(np.uint8)` of run_edit.py:267 — addressed by element strides (image, row, column, channel), so the
 * pipeline's [3, F, H, W] video tensor is read in place.  dst: uint8 [n, Hd, Wd, C] contiguous.
 *   replaces: VaeImageProcessor LANCZOS resize (pipeline_pose2vid_long_edit_bkfill_roiclip.py:424-457), the (224, 224)
 *   bicubic resize before CLIPImageProcessor (:379-384), res_image_pil.resize((pad_w, pad_h)) (run_edit.py:268-269). */
int mimo_resample_pass_u8(const void* src, int src_is_f32, int64_t src_stride_n, int64_t src_stride_y,
                          int64_t src_stride_x, int64_t src_stride_c, void* dst, int n, int Hd, int Wd, int C,
                          const int* bounds, const int* coeffs, int ksize, int horizontal, void* stream);
/* uint8 [npix, C] -> half16 (dtype MIMO_F32: fp32) tokens [npix, Cpad] = x / 255 (fp32), optionally 2 x - 1; channels >= C zero
 *   (VaeImageProcessor.preprocess as configured at pipeline_...roiclip.py:73-80). */
int mimo_u8_to_tokens(int dtype, const void* src, int64_t npix, int C, int Cpad, int two_x_minus_1, void* dst,
                      void* stream);
/* uint8 [n, HW, C] -> fp32 planar [n, C, HW] = (x * rescale - mean[c]) / std[c]  (CLIPImageProcessor rescale + normalize) */
int mimo_u8_to_planar_f32(const void* src, int n, int64_t HW, int C, float rescale, const float* mean,
                          const float* stdv, float* dst, void* stream);
/* The per-frame compositing of run_edit.py:253-304 in one pass over the full-resolution frame:
 *   canvas = white; paste crop[top : pad_h - bottom, left : pad_w - right] at (w_min, h_min)
 *   res = canvas * mask_full + bk * (1 - mask_full)          float32; mask_full = 0 outside mask placed at (h_min, w_min)
 *   res = res * (1 - occ / 255.0) + vid * (occ / 255.0)      float64, when occ != NULL (channel 0 of occ)
 *   res = prev * (1 - factor) + res * factor                 float64 (a float32 res * factor stays float32), when prev != NULL
 *   out = (uint8) res                                        truncation, as ndarray.astype(np.uint8)
 * Every operation is separately rounded in numpy's width and order: the uint8 result is the reference's. */
typedef struct mimo_composite_params {
  const void* crop;   /* uint8 [pad_h, pad_w, 3]: the generated frame resized to the padded clip size */
  const float* mask;  /* fp32 [mh, mw]: the (already resized) alpha mask of tools/util.py:397-447 */
  const void* bk;     /* uint8 [H, W, 3] inpainted background frame */
  const void* occ;    /* uint8 [H, W, 3] occluder mask frame, or NULL */
  const void* vid;    /* uint8 [H, W, 3] original frame (needed with occ) */
  const void* prev;   /* uint8 [H, W, 3] result of the previous clip for this frame (overlap cross-fade), or NULL */
  void* out;          /* uint8 [H, W, 3]; may alias prev */
  double factor;      /* (i - start_i + 1) / (overlay + 1) */
  int pad_h, pad_w, top, bottom, left, right, w_min, h_min, mh, mw, H, W;
} mimo_composite_params;
int mimo_composite_frame(const mimo_composite_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIMO_HIP_H */
