// gemm_conv.hip — the MFMA workhorse of the denoising path: one LDS-tiled kernel that is
// either a dense GEMM  out[M,N] = A[M,K].W[N,K]^T  or a channels-last implicit-GEMM
// convolution (3x3 / 1x1, stride 1|2, optional nearest-upsample gather, optional extra 1x1
// tap over a second tensor = fused ResBlock shortcut), with a fused epilogue
// (bias, per-batch bias = time embedding / collapsed cross-attention, SiLU, GEGLU,
// residual add, fp32|half store).
//
// Reference ops replaced: see include/mimo_hip.h (mimo_gemm / mimo_conv2d).
//
// Tiling (gfx950): block = WM x WN waves; block tile (16*MT*WM) x (16*NR*WN) x 64; each wave owns
// (16*MT) x (16*NR) as MT x NR MFMA 16x16x32 tiles, fp32 accumulators in VGPRs.
// Operands go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`): hardware range checking
// turns padding taps / ragged edges into zero fill, the XOR swizzle that makes the
// ds_read_b128 fragment reads bank-conflict-free is applied on the SOURCE address (the DMA
// writes lane-linear), and an NSTAGE-deep LDS ring keeps NSTAGE-1 K-tiles in flight under
// the MFMAs with ONE barrier and one counted `s_waitcnt vmcnt` per K-tile.  Configurations
// (picked per shape in launch_nr): S 2x2 waves, 128-row tile, 2 blocks/CU; L 4x2, 3-deep ring;
// XL 4x4 = 16 waves 256 x 256 (GEGLU); XL8 2x4 = 8 waves with a 256-register budget,
// (256|192|128) x 320; split-K over gridDim.y for long reductions over few tiles;
// gemm_dense_persist_kernel = the dense case as persistent blocks with cross-tile prefetch.
// The MFMA is issued "swapped" (W fragment as the A operand) so that every lane ends up
// with 4 consecutive output columns of one row -> 16-byte epilogue loads/stores.
#include <stdlib.h>

#include "common.hip.h"
#include "gemm_stream.hip.h"
#include "thinconv.hip.h"

namespace {

constexpr int BK = 64;

// internal launch flags (upper half of GemmArgs::flags; the public MIMO_EPI_* bits live in the lower half)
constexpr unsigned F_STAGGER = 0x40000u;    // waves sharing a SIMD issue their DMAs at different points of the K-tile
constexpr unsigned F_TAP_INNER = 0x80000u;  // convolution K order: channel chunk outer, tap inner
#ifdef MIMO_TUNE  // timing experiments of tools/microbench.py (results are wrong): never compiled into the shipped library
#define MIMO_ABLATE(g, bit) (((g).flags & (bit)) != 0u)
#else
#define MIMO_ABLATE(g, bit) false
#endif
constexpr unsigned F_ABL_NO_DMA = 0x10000u, F_ABL_NO_MFMA = 0x20000u, F_ABL_NO_GELU = 0x100000u;

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* A2;
  const uint16_t* W;
  void* out;
  const float* bias;
  const float* img_bias;
  const void* res;
  int64_t lda, ldw, ldo, ldr, M, rows_per_img, ldib;
  unsigned a_bytes, a2_bytes, w_bytes;  // buffer-descriptor extents (each operand < 2 GiB)
  int N, K;
  float out_scale;
  unsigned flags;
  int tiles_n;
  unsigned ntiles;     // persistent kernels: number of output tiles; gemm8_kernel: row panels per tile-order group (0 | 1: row-major)
  void* ws;            // split-K workspace (fp32 partial tiles), may be null
  size_t ws_bytes;
  // conv geometry
  int Hin, Win, Cin, Hout, Wout, ks, stride, pad_t, pad_l, Hup, Wup, Cin2;
  float sh, sw;
  int chunks1, chunks2, nkt;
  // optional fused side outputs (mimo_epilogue_ext)
  float* colstats;       // [M/32][2][N]: per 32-row slab and column (mean, centred sum of squares) of the stored values
  const float* ln_gamma; // LayerNorm over the N columns of every output row (needs N == tile width): gamma, beta [N]
  const float* ln_beta;
  const float* ln_pe;    // optional additive table [pe_frames][N], row = (m / ln_rows_per_frame) % ln_pe_frames
  void* ln_out;          // half16 [M, N] (ld = N)
  float ln_eps;
  int64_t ln_rows_per_frame;
  int ln_pe_frames;
  // LayerNorm folded into the consuming GEMM (mimo_epilogue_ext row_half / row_stats / a_row_stats / a_colsum)
  void* row_half;            // producer: half16 [M, N] copy of the stored rows (ld = N)
  float* row_stats;          // producer: [M][N / slot width][2] = (sum, sum of squares) of the ROUNDED values of each column slot
  const float* a_row_stats;  // consumer: the row_stats of A's rows ([M][a_slots][2]); v = rstd * (acc - mean * a_colsum[n])
  const float* a_colsum;     // consumer: [N] = sum over k of W[n, k]
  int a_slots;
  float a_eps;
  unsigned long long* dbg;  // MIMO_TUNE builds: cycle-counter trace of block 0 (null otherwise)
};

// Trace points (tune build, MIMO_GEMM_TRACE=1): thread 0 of block 0 appends (tag << 56 | s_memtime) to g.dbg.
#ifdef MIMO_TUNE
#define MIMO_TRACE(g, idx, tag)                                                                                    \
  do {                                                                                                             \
    if ((g).dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (idx) < 4000u)                        \
      (g).dbg[(idx)++] = ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); \
  } while (0)
#define MIMO_TRACE_REAL(g, idx, tag)                                                                                  \
  do {                                                                                                                \
    if ((g).dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (idx) < 4000u)                           \
      (g).dbg[(idx)++] = ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull); \
  } while (0)
#else
#define MIMO_TRACE(g, idx, tag) do { (void)(idx); } while (0)
#define MIMO_TRACE_REAL(g, idx, tag) do { (void)(idx); } while (0)
#endif

template <int V>
struct IC {
  static constexpr int value = V;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// ---- epilogue of one output tile: lane holds out[m = M0 + .. + li][n = N0 + .. + 4*lg + r], r = 0..3 ----
// PAIR_IMGB: half output with a per-image bias also takes the paired 16-byte-store form (its bias rows are fetched one row
// pair ahead, in front of the previous pair's stores) — the QKV projections with a folded LayerNorm + positional table
template <int DT, int NR, int MT, int BM, bool PAIR = true, bool PAIR_IMGB = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, f32x4 (&acc)[NR][MT], const int64_t M0, const int N0,
                                              const int wm, const int wn, const int lg, const int li,
                                              const unsigned ysplit) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  // Branch-free per element: every operand goes through a buffer descriptor clipped to THIS tile's valid rows, so
  // ragged rows / columns are hardware range checks (loads return 0, stores are dropped) and all loads of one
  // 16-column group are issued back to back before the math.  The variants (GEGLU | residual x per-image bias) are
  // separate straight-line instantiations selected by uniform branches.
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32;
  const bool res_f32 = g.flags & MIMO_EPI_RES_F32;
  const bool do_silu = g.flags & MIMO_EPI_SILU;
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const int n_out = geglu ? g.N / 2 : g.N;
  const unsigned esz_o = out_f32 ? 4u : 2u, esz_r = res_f32 ? 4u : 2u;
  const bool pair16 = (n_out & 7) == 0;  // half output in 16-byte stores (lane exchange below)
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.out + ((int64_t)ysplit * g.M + M0) * g.ldo * esz_o, 0,
      (int)(((rows_valid - 1) * g.ldo + n_out) * esz_o), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      (char*)const_cast<void*>(g.res) + M0 * g.ldr * esz_r, 0,
      g.res ? (int)(((rows_valid - 1) * g.ldr + n_out) * esz_r) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  const int64_t nimg = g.img_bias ? (g.M + g.rows_per_img - 1) / g.rows_per_img : 0;
  const __amdgpu_buffer_rsrc_t r_imgb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.img_bias), 0, g.img_bias ? (int)(((nimg - 1) * g.ldib + g.N) * 4) : 0, 0x00020000);
  const int row0 = wm * 16 * MT + li;       // tile-local row of mi = 0
  const int col0 = wn * 16 * NR + 4 * lg;   // tile-local column of ni = 0, r = 0
  auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MIMO_LD_AUX));
  };

  auto epilogue = [&](auto has_res_c, auto has_imgb_c, auto geglu_c) {
    constexpr bool HAS_RES = decltype(has_res_c)::value != 0;
    constexpr bool HAS_IMGB = decltype(has_imgb_c)::value != 0;
    constexpr bool GEGLU = decltype(geglu_c)::value != 0;
    // per-column-group constants: bias (for GEGLU bv[ni + 1] is the gate bias), column offsets, validity
    f32x4 bv[NR];
    bool col_ok[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int n = N0 + col0 + ni * 16;
      col_ok[ni] = n < g.N;
      bv[ni] = ld4(r_bias, col_ok[ni] ? (unsigned)n * 4u : OOB);
    }
    // rows outer, column groups inner: consecutive stores of a lane fill one output row left to right
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const unsigned row = (unsigned)(row0 + mi * 16);
      f32x4 ib[NR], rr[NR];
      if (HAS_IMGB) {  // host guarantees M < 2^31 when a per-image bias is given
        const unsigned img_off = ((unsigned)(M0 + row) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u;
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
          ib[ni] = ld4(r_imgb, col_ok[ni] ? img_off + (unsigned)(N0 + col0 + ni * 16) * 4u : OOB);
      }
      if (HAS_RES) {
        if (res_f32) {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni)
            rr[ni] = ld4(r_res, col_ok[ni] ? (row * (unsigned)g.ldr + (unsigned)(N0 + col0 + ni * 16)) * 4u : OOB);
        } else {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) {
            const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(
                r_res, col_ok[ni] ? (row * (unsigned)g.ldr + (unsigned)(N0 + col0 + ni * 16)) * 2u : OOB, 0, MIMO_LD_AUX);
            rr[ni] = (f32x4){HT<DT>::to_f((uint16_t)(h.x & 0xffffu)), HT<DT>::to_f((uint16_t)(h.x >> 16)),
                             HT<DT>::to_f((uint16_t)(h.y & 0xffffu)), HT<DT>::to_f((uint16_t)(h.y >> 16))};
          }
        }
      }
#pragma unroll
      for (int ni = 0; ni < NR; ni += (GEGLU ? 2 : 1)) {
        if (GEGLU && ni + 1 >= NR) break;
        f32x4 v = acc[ni][mi] + bv[ni];
        if (HAS_IMGB) v += ib[ni];
        if (GEGLU) {
          // gate tile = the next 16 packed columns; identical lane mapping
          const int ng = (ni + 1 < NR) ? ni + 1 : ni;
          const f32x4 gt = acc[ng][mi] + bv[ng];
          v *= MIMO_ABLATE(g, F_ABL_NO_GELU) ? gt : gelu_erf_4(gt);
        }
        if (do_silu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
        }
        if (HAS_RES) v += rr[ni];
        v *= g.out_scale;
        const int no = GEGLU ? (N0 + wn * 16 * NR + ni * 16) / 2 + 4 * lg : N0 + col0 + ni * 16;  // output column
        const unsigned ooff = col_ok[ni] ? (row * (unsigned)g.ldo + (unsigned)no) * esz_o : OOB;
        if (out_f32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, ooff, 0, MIMO_ST_AUX);
        } else {
          u32x2 o;
          o.x = pack2<DT>(v[0], v[1]);
          o.y = pack2<DT>(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, r_out, ooff, 0, MIMO_ST_AUX);
        }
      }
    }
  };
  // half output in 16-byte stores.  A lane holds 4 consecutive columns of a row; the two rows of a pair of MFMA tiles are
  // exchanged between neighbouring 16-lane rows (v_permlane16_swap) so that every lane stores 8 consecutive columns of
  // ONE row: the store path retires about one wave-instruction per 70 cycles whatever its width, and 8-byte stores
  // capped the short-K GEMMs near 7 B/clk per CU.  Operands are fetched per 16-column group, just in time (registers).
  auto epilogue_pair = [&](auto has_res_c, auto has_imgb_c, auto geglu_c) {
    constexpr bool HAS_RES = decltype(has_res_c)::value != 0;
    constexpr bool HAS_IMGB = decltype(has_imgb_c)::value != 0;
    constexpr bool GEGLU = decltype(geglu_c)::value != 0;
    static_assert(MT % 2 == 0, "rows are processed in pairs of MFMA tiles");
    // every bias load is issued before the first store: vmcnt retires in order, a load behind a store would wait for
    // the store's acknowledgement
    f32x4 bv[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int n = N0 + col0 + ni * 16;
      bv[ni] = ld4(r_bias, n < g.N ? (unsigned)n * 4u : OOB);
    }
    // the per-image bias rows of a row pair are fetched while the previous pair is being stored (a load issued behind a
    // store would wait for the store's acknowledgement: vmcnt retires in order)
    auto ib_load = [&](int mi, int ni) -> f32x4 {
      const unsigned row = (unsigned)(row0 + mi * 16);
      const int n = N0 + col0 + ni * 16;
      return ld4(r_imgb, n < g.N ? ((unsigned)(M0 + row) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u + (unsigned)n * 4u : OOB);
    };
    [[maybe_unused]] f32x4 ibc[2][NR], ibn[2][NR];
    if constexpr (HAS_IMGB) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) ibc[u][ni] = ib_load(u, ni);
    }
    // rows outer, column groups inner: consecutive stores fill the two rows of the pair left to right
#pragma unroll
    for (int mi0 = 0; mi0 < MT; mi0 += 2) {
      if constexpr (HAS_IMGB) {
        if (mi0 + 2 < MT) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ni = 0; ni < NR; ++ni) ibn[u][ni] = ib_load(mi0 + 2 + u, ni);
        }
      }
#pragma unroll
      for (int ni = 0; ni < NR; ni += (GEGLU ? 2 : 1)) {
        if (GEGLU && ni + 1 >= NR) break;
        const int n = N0 + col0 + ni * 16;
        const bool okc = n < g.N;
        const int nt = GEGLU ? (N0 + wn * 16 * NR + ni * 16) / 2 : N0 + wn * 16 * NR + ni * 16;  // first output column of the tile
        const unsigned c8 = (unsigned)(nt + 4 * (lg & ~1));
        f32x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int mi = mi0 + u;
          const unsigned row = (unsigned)(row0 + mi * 16);
          v[u] = acc[ni][mi] + bv[ni];
          if (HAS_IMGB) v[u] += ibc[u][ni];
          if (GEGLU) {
            const int ng = ni + 1 < NR ? ni + 1 : ni;
            const f32x4 gt = acc[ng][mi] + bv[ng];
            v[u] *= MIMO_ABLATE(g, F_ABL_NO_GELU) ? gt : gelu_erf_4(gt);
          }
          if (do_silu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[u][r] = silu_f(v[u][r]);
          }
          if (HAS_RES) {
            if (res_f32) {
              v[u] += ld4(r_res, okc ? (row * (unsigned)g.ldr + (unsigned)n) * 4u : OOB);
            } else {
              const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(r_res, okc ? (row * (unsigned)g.ldr + (unsigned)n) * 2u : OOB, 0, MIMO_LD_AUX);
              v[u] += (f32x4){HT<DT>::to_f((uint16_t)(h.x & 0xffffu)), HT<DT>::to_f((uint16_t)(h.x >> 16)),
                              HT<DT>::to_f((uint16_t)(h.y & 0xffffu)), HT<DT>::to_f((uint16_t)(h.y >> 16))};
            }
          }
          v[u] *= g.out_scale;
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(pack2<DT>(v[0][0], v[0][1]), pack2<DT>(v[1][0], v[1][1]), false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pack2<DT>(v[0][2], v[0][3]), pack2<DT>(v[1][2], v[1][3]), false, false);
        const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
        // even 16-lane rows: row mi0, columns 4 lg .. 4 lg + 7; odd rows: row mi0 + 1, columns 4 (lg - 1) ..
        const unsigned row = (unsigned)(row0 + (mi0 + (lg & 1)) * 16);
        const unsigned ooff = (int)c8 < n_out ? (row * (unsigned)g.ldo + c8) * 2u : OOB;  // n_out % 8 == 0
        __builtin_amdgcn_raw_buffer_store_b128(o, r_out, ooff, 0, MIMO_ST_AUX);
      }
      if constexpr (HAS_IMGB) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) ibc[u][ni] = ibn[u][ni];
      }
    }
  };
  // (GEGLU and the 16-wave tiles keep the per-row form: 128 registers per wave, the paired form spills there)
  // Variants with a residual or per-image bias also keep the per-row form: it issues a row's loads ahead of the
  // previous row's stores, the paired form would fetch them just in time behind its own stores (measured slower).
  if constexpr (PAIR) {
    if (!out_f32 && pair16 && !geglu && !g.res && !g.img_bias) {
      epilogue_pair(IC<0>{}, IC<0>{}, IC<0>{});
      return;
    }
    if constexpr (PAIR_IMGB) {
      if (!out_f32 && pair16 && !geglu && !g.res && g.img_bias) {
        epilogue_pair(IC<0>{}, IC<1>{}, IC<0>{});
        return;
      }
    }
  }
  if (geglu) epilogue(IC<0>{}, IC<0>{}, IC<1>{});
  else if (g.res && g.img_bias) epilogue(IC<1>{}, IC<1>{}, IC<0>{});
  else if (g.res) epilogue(IC<1>{}, IC<0>{}, IC<0>{});
  else if (g.img_bias) epilogue(IC<0>{}, IC<1>{}, IC<0>{});
  else epilogue(IC<0>{}, IC<0>{}, IC<0>{});
}


// ---- LayerNorm folded into the CONSUMING GEMM (round 5; C = 640 / 1280, where no tile holds a whole row) ----
// LayerNorm(x) . W^T = rstd * (x . (gamma o W)^T - mean * colsum(gamma o W)) + beta . W^T: the consumer multiplies the RAW rows
// (rounded to half: `row_half`) by the gamma-scaled weight and applies mean / rstd per row in its epilogue, so the
// normalised tensor is never written and the fp32 tensor never re-read.  The producer of x leaves, next to its fp32 result,
// the half copy and per row the (sum, sum of squares) of the rounded values of every column slot; a slot is the 16 NR
// columns one wave owns (a function of N only: row_slot_width), so the partials and their fixed summation order are the
// same whatever tile height or kernel the row count selects.
template <int DT, int NR, int MT, int BM>
__device__ __forceinline__ void tile_epilogue_rowside(const GemmArgs& g, f32x4 (&acc)[NR][MT], const int64_t M0, const int N0,
                                                      const int wm, const int wn, const int lg, const int li) {
  // The producer's whole epilogue, two MFMA row tiles at a time (so the accumulators retire as they are stored):
  // v = acc + bias (+ per-image bias) (+ residual), * out_scale -> fp32 `out`; half(v) -> `row_half` in 16-byte stores through
  // the lane exchange of tile_epilogue; (sum, sum of squares) of the rounded values of this wave's 16 NR columns -> `row_stats`.
  static_assert(MT % 2 == 0, "rows are stored in pairs of MFMA tiles");
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const bool res_f32 = g.flags & MIMO_EPI_RES_F32;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const unsigned esz_r = res_f32 ? 4u : 2u;
  const int slots = g.N / (16 * NR);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.out + M0 * g.ldo * 4, 0, (int)(((rows_valid - 1) * g.ldo + g.N) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      (char*)const_cast<void*>(g.res) + M0 * g.ldr * esz_r, 0,
      g.res ? (int)(((rows_valid - 1) * g.ldr + g.N) * esz_r) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  const int64_t nimg = g.img_bias ? (g.M + g.rows_per_img - 1) / g.rows_per_img : 0;
  const __amdgpu_buffer_rsrc_t r_imgb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.img_bias), 0, g.img_bias ? (int)(((nimg - 1) * g.ldib + g.N) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_h = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.row_half + M0 * g.N * 2, 0, (int)(rows_valid * g.N * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.row_stats + M0 * slots * 2), 0, (int)(rows_valid * slots * 8), 0x00020000);
  const int row0 = wm * 16 * MT + li;
  const int colw = N0 + wn * 16 * NR;        // first column of this wave's slot (N % (16 NR) == 0: every column is valid)
  const unsigned slot = (unsigned)(colw / (16 * NR));
  auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MIMO_LD_AUX));
  };
#pragma unroll
  for (int sp = 0; sp < MT / 2; ++sp) {
    const unsigned rowa = (unsigned)(row0 + (2 * sp) * 16), rowb = rowa + 16u;
    unsigned imga = 0, imgb = 0;
    if (g.img_bias) {
      imga = ((unsigned)(M0 + rowa) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u;
      imgb = ((unsigned)(M0 + rowb) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u;
    }
    // 1: every load of the row pair back to back (vmcnt retires in order: a load behind a store waits for the store)
    f32x4 va[NR], vb[NR], ra[NR], rb[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const unsigned n = (unsigned)(colw + ni * 16 + 4 * lg);
      const f32x4 bv = ld4(r_bias, n * 4u);
      va[ni] = acc[ni][2 * sp] + bv;
      vb[ni] = acc[ni][2 * sp + 1] + bv;
      if (g.img_bias) {
        va[ni] += ld4(r_imgb, imga + n * 4u);
        vb[ni] += ld4(r_imgb, imgb + n * 4u);
      }
      if (g.res) {
        if (res_f32) {
          ra[ni] = ld4(r_res, (rowa * (unsigned)g.ldr + n) * 4u);
          rb[ni] = ld4(r_res, (rowb * (unsigned)g.ldr + n) * 4u);
        } else {
          const u32x2 ha = __builtin_amdgcn_raw_buffer_load_b64(r_res, (rowa * (unsigned)g.ldr + n) * 2u, 0, MIMO_LD_AUX);
          const u32x2 hb = __builtin_amdgcn_raw_buffer_load_b64(r_res, (rowb * (unsigned)g.ldr + n) * 2u, 0, MIMO_LD_AUX);
          ra[ni] = (f32x4){HT<DT>::to_f((uint16_t)(ha.x & 0xffffu)), HT<DT>::to_f((uint16_t)(ha.x >> 16)),
                           HT<DT>::to_f((uint16_t)(ha.y & 0xffffu)), HT<DT>::to_f((uint16_t)(ha.y >> 16))};
          rb[ni] = (f32x4){HT<DT>::to_f((uint16_t)(hb.x & 0xffffu)), HT<DT>::to_f((uint16_t)(hb.x >> 16)),
                           HT<DT>::to_f((uint16_t)(hb.y & 0xffffu)), HT<DT>::to_f((uint16_t)(hb.y >> 16))};
        }
      }
    }
    // 2: finish, store fp32, round, store the half copy, accumulate the slot's sums
    float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const unsigned n = (unsigned)(colw + ni * 16 + 4 * lg);
      if (g.res) { va[ni] += ra[ni]; vb[ni] += rb[ni]; }
      va[ni] *= g.out_scale;
      vb[ni] *= g.out_scale;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, va[ni]), r_out, (rowa * (unsigned)g.ldo + n) * 4u, 0, MIMO_ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb[ni]), r_out, (rowb * (unsigned)g.ldo + n) * 4u, 0, MIMO_ST_AUX);
      const unsigned la = pack2<DT>(va[ni][0], va[ni][1]), ha = pack2<DT>(va[ni][2], va[ni][3]);
      const unsigned lb = pack2<DT>(vb[ni][0], vb[ni][1]), hb = pack2<DT>(vb[ni][2], vb[ni][3]);
      {
        const float h0 = HT<DT>::to_f((uint16_t)(la & 0xffffu)), h1 = HT<DT>::to_f((uint16_t)(la >> 16));
        const float h2 = HT<DT>::to_f((uint16_t)(ha & 0xffffu)), h3 = HT<DT>::to_f((uint16_t)(ha >> 16));
        s1a += (h0 + h1) + (h2 + h3);
        s2a += fmaf(h0, h0, h1 * h1) + fmaf(h2, h2, h3 * h3);
      }
      {
        const float h0 = HT<DT>::to_f((uint16_t)(lb & 0xffffu)), h1 = HT<DT>::to_f((uint16_t)(lb >> 16));
        const float h2 = HT<DT>::to_f((uint16_t)(hb & 0xffffu)), h3 = HT<DT>::to_f((uint16_t)(hb >> 16));
        s1b += (h0 + h1) + (h2 + h3);
        s2b += fmaf(h0, h0, h1 * h1) + fmaf(h2, h2, h3 * h3);
      }
      // even 16-lane rows end up with row a, columns 4 lg .. 4 lg + 7; odd ones with row b, columns 4 (lg - 1) ..
      const auto sx = __builtin_amdgcn_permlane16_swap(la, lb, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(ha, hb, false, false);
      const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
      const unsigned row = (lg & 1) ? rowb : rowa;
      const unsigned c8 = (unsigned)(colw + ni * 16 + 4 * (lg & ~1));
      __builtin_amdgcn_raw_buffer_store_b128(o, r_h, (row * (unsigned)g.N + c8) * 2u, 0, MIMO_ST_AUX);
    }
    // 3: the four lanes li, li + 16, li + 32, li + 48 hold the four column quads of a row: (q0 + q1) + (q2 + q3) everywhere
    s1a += __shfl_xor(s1a, 16); s2a += __shfl_xor(s2a, 16); s1b += __shfl_xor(s1b, 16); s2b += __shfl_xor(s2b, 16);
    s1a += __shfl_xor(s1a, 32); s2a += __shfl_xor(s2a, 32); s1b += __shfl_xor(s1b, 32); s2b += __shfl_xor(s2b, 32);
    u32x2 st;
    st.x = __builtin_bit_cast(unsigned, s1a); st.y = __builtin_bit_cast(unsigned, s2a);
    __builtin_amdgcn_raw_buffer_store_b64(st, r_s, lg == 0 ? (rowa * (unsigned)slots + slot) * 8u : OOB, 0, 0);
    st.x = __builtin_bit_cast(unsigned, s1b); st.y = __builtin_bit_cast(unsigned, s2b);
    __builtin_amdgcn_raw_buffer_store_b64(st, r_s, lg == 0 ? (rowb * (unsigned)slots + slot) * 8u : OOB, 0, 0);
  }
}

// the consumer half: acc <- rstd[m] * (acc - mean[m] * colsum[n]) in front of the ordinary epilogue (bias = beta . W^T + b,
// GEGLU, ...).  Nothing of it may wait on memory in the epilogue (a tile lives ~25 us; fetched there — four lanes of every wave
// of a block row each loading a row's partials — the statistics cost 45-75 us per launch, and even one exposed L2 round
// trip + two barriers per tile cost 10-20 us): a row's partial sums (K columns in a_slots <= 20 slots) are fetched by ONE
// thread per row when the block starts, summed left to right while the prologue's DMAs are in flight, and (mean, rstd) of
// the BM rows and the colsum of the BN columns wait in a small LDS region of their own (3 KB) for the epilogue.
constexpr int ROW_STAT_LOADS = 10;   // 16-byte loads = two slots each

__device__ __forceinline__ void row_stats_issue(const GemmArgs& g, const int64_t M0, const int N0, const int BM, const int BN, const int tid,
                                                u32x4 (&q)[ROW_STAT_LOADS], u32x4& cq) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const unsigned slots = (unsigned)g.a_slots;
  const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.a_row_stats ? g.a_row_stats + M0 * slots * 2 : nullptr), 0,
      g.a_row_stats ? (int)(rows_valid * slots * 8) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.a_colsum), 0, g.a_colsum ? g.N * 4 : 0, 0x00020000);
  const unsigned base = (unsigned)tid * slots * 8u;
#pragma unroll
  for (int p = 0; p < ROW_STAT_LOADS; ++p)
    q[p] = __builtin_amdgcn_raw_buffer_load_b128(r_s, (tid < BM && 2u * (unsigned)p < slots) ? base + (unsigned)p * 16u : OOB, 0, 0);
  const int n = N0 + 4 * tid;   // thread tid < BN / 4: four columns of the tile
  cq = __builtin_amdgcn_raw_buffer_load_b128(r_c, (4 * tid < BN && n < g.N) ? (unsigned)n * 4u : OOB, 0, 0);
}

// thread tid < BM: (mean, rstd) of row tid -> mr[2 tid ..]; thread tid < BN / 4: colsum of columns 4 tid .. -> cs[4 tid ..]
__device__ __forceinline__ void row_stats_finish(const GemmArgs& g, const int BM, const int BN, const int tid,
                                                 const u32x4 (&q)[ROW_STAT_LOADS], const u32x4& cq, float* mr, float* cs) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int p = 0; p < ROW_STAT_LOADS; ++p) {   // (slots beyond a_slots were out of range: + 0)
    const f32x4 v = __builtin_bit_cast(f32x4, q[p]);
    s1 += v[0]; s2 += v[1];
    s1 += v[2]; s2 += v[3];
  }
  const float inv_k = 1.0f / (float)g.K;
  const float mean = s1 * inv_k;
  const float var = fmaxf(fmaf(-mean, mean, s2 * inv_k), 0.f);
  if (tid < BM) {
    mr[2 * tid] = mean;
    mr[2 * tid + 1] = rsqrtf(var + g.a_eps);
  }
  if (4 * tid < BN) *reinterpret_cast<u32x4*>(cs + 4 * tid) = cq;
}

template <int NR, int MT>
__device__ __forceinline__ void tile_row_affine(f32x4 (&acc)[NR][MT], const int wm, const int wn, const int lg, const int li,
                                                const float* mr, const float* cs) {
  const int row0 = wm * 16 * MT + li;
  const int col0 = wn * 16 * NR + 4 * lg;
  float mean[MT], rstd[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    mean[mi] = mr[2 * (row0 + mi * 16)];
    rstd[mi] = mr[2 * (row0 + mi * 16) + 1];
  }
#pragma unroll
  for (int ni = 0; ni < NR; ++ni) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(cs + col0 + ni * 16);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (acc[ni][mi] - mean[mi] * c) * rstd[mi];
  }
}

// sum over the 16 lanes of a DPP row (= the 16 rows `li` of one MFMA tile that share a column chunk): four
// v_add_f32_dpp, every lane ends up with the total, fixed order (deterministic)
__device__ __forceinline__ float row16_sum(float x) {
  auto dpp = [](float v, auto ctrl_c) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_c)::value, 0xf, 0xf, true));
  };
  x += dpp(x, IC<0xB1>{});   // quad_perm [1,0,3,2]
  x += dpp(x, IC<0x4E>{});   // quad_perm [2,3,0,1]
  x += dpp(x, IC<0x141>{});  // row_half_mirror
  x += dpp(x, IC<0x140>{});  // row_mirror
  return x;
}

// ---- epilogue variant that also emits GroupNorm column statistics ----
// GroupNorm statistics of the tensor this launch writes are produced HERE, from the fp32 values in registers,
// instead of by a second pass over HBM: for every 32-row slab (two MFMA row tiles) and every column the epilogue
// writes (mean, sum of squared deviations from that mean) — an exact two-pass computation inside the slab — and
// mimo_group_norm_stats_cols merges the slabs of an image and the columns of a group (Chan's parallel update, in
// double, fixed order).  Loop order: slab outer, column group inner, so only two row tiles of values are live.
template <int DT, int NR, int MT, int BM>
__device__ __forceinline__ void tile_epilogue_stats(const GemmArgs& g, f32x4 (&acc)[NR][MT], const int64_t M0, const int N0,
                                                    const int wm, const int wn, const int lg, const int li) {
  static_assert(MT % 2 == 0, "32-row slabs");
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32;
  const bool res_f32 = g.flags & MIMO_EPI_RES_F32;
  const bool do_silu = g.flags & MIMO_EPI_SILU;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const unsigned esz_o = out_f32 ? 4u : 2u, esz_r = res_f32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.out + M0 * g.ldo * esz_o, 0, (int)(((rows_valid - 1) * g.ldo + g.N) * esz_o), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      (char*)const_cast<void*>(g.res) + M0 * g.ldr * esz_r, 0,
      g.res ? (int)(((rows_valid - 1) * g.ldr + g.N) * esz_r) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  const int64_t nimg = g.img_bias ? (g.M + g.rows_per_img - 1) / g.rows_per_img : 0;
  const __amdgpu_buffer_rsrc_t r_imgb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.img_bias), 0, g.img_bias ? (int)(((nimg - 1) * g.ldib + g.N) * 4) : 0, 0x00020000);
  // colstats rows of this tile: slab index (M0 + row) / 32, two planes of N floats each
  const int64_t slab0 = M0 >> 5;
  const int64_t slabs_valid = (rows_valid + 31) >> 5;
  const __amdgpu_buffer_rsrc_t r_cs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(g.colstats + slab0 * 2 * g.N), 0, (int)(slabs_valid * 2 * g.N * 4), 0x00020000);
  const int row0 = wm * 16 * MT + li;
  const int col0 = wn * 16 * NR + 4 * lg;
  auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MIMO_LD_AUX));
  };
#pragma unroll
  for (int sp = 0; sp < MT / 2; ++sp) {
    const unsigned rowa = (unsigned)(row0 + (2 * sp) * 16), rowb = rowa + 16u;
    unsigned imga = 0, imgb = 0;
    if (g.img_bias) {
      imga = ((unsigned)(M0 + rowa) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u;
      imgb = ((unsigned)(M0 + rowb) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u;
    }
    const unsigned slab_local = (unsigned)((wm * 16 * MT) >> 5) + (unsigned)sp;
    // 1: every load of the slab back to back (interleaved stores would serialise them: vmcnt retires in order)
    f32x4 va[NR], vb[NR], ra[NR], rb[NR];
    bool okc[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int n = N0 + col0 + ni * 16;
      okc[ni] = n < g.N;
      const f32x4 bv = ld4(r_bias, okc[ni] ? (unsigned)n * 4u : OOB);
      va[ni] = acc[ni][2 * sp] + bv;
      vb[ni] = acc[ni][2 * sp + 1] + bv;
      if (g.img_bias) {
        va[ni] += ld4(r_imgb, okc[ni] ? imga + (unsigned)n * 4u : OOB);
        vb[ni] += ld4(r_imgb, okc[ni] ? imgb + (unsigned)n * 4u : OOB);
      }
      if (g.res) {
        if (res_f32) {
          ra[ni] = ld4(r_res, okc[ni] ? (rowa * (unsigned)g.ldr + (unsigned)n) * 4u : OOB);
          rb[ni] = ld4(r_res, okc[ni] ? (rowb * (unsigned)g.ldr + (unsigned)n) * 4u : OOB);
        } else {
          const u32x2 ha = __builtin_amdgcn_raw_buffer_load_b64(r_res, okc[ni] ? (rowa * (unsigned)g.ldr + (unsigned)n) * 2u : OOB, 0, MIMO_LD_AUX);
          const u32x2 hb = __builtin_amdgcn_raw_buffer_load_b64(r_res, okc[ni] ? (rowb * (unsigned)g.ldr + (unsigned)n) * 2u : OOB, 0, MIMO_LD_AUX);
          ra[ni] = (f32x4){HT<DT>::to_f((uint16_t)(ha.x & 0xffffu)), HT<DT>::to_f((uint16_t)(ha.x >> 16)),
                           HT<DT>::to_f((uint16_t)(ha.y & 0xffffu)), HT<DT>::to_f((uint16_t)(ha.y >> 16))};
          rb[ni] = (f32x4){HT<DT>::to_f((uint16_t)(hb.x & 0xffffu)), HT<DT>::to_f((uint16_t)(hb.x >> 16)),
                           HT<DT>::to_f((uint16_t)(hb.y & 0xffffu)), HT<DT>::to_f((uint16_t)(hb.y >> 16))};
        }
      }
    }
    // 2: finish the values and store them
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int n = N0 + col0 + ni * 16;
      if (do_silu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { va[ni][r] = silu_f(va[ni][r]); vb[ni][r] = silu_f(vb[ni][r]); }
      }
      if (g.res) { va[ni] += ra[ni]; vb[ni] += rb[ni]; }
      va[ni] *= g.out_scale;
      vb[ni] *= g.out_scale;
      const unsigned oa = okc[ni] ? (rowa * (unsigned)g.ldo + (unsigned)n) * esz_o : OOB;
      const unsigned ob = okc[ni] ? (rowb * (unsigned)g.ldo + (unsigned)n) * esz_o : OOB;
      if (out_f32) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, va[ni]), r_out, oa, 0, MIMO_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb[ni]), r_out, ob, 0, MIMO_ST_AUX);
      } else {
        u32x2 o;
        o.x = pack2<DT>(va[ni][0], va[ni][1]); o.y = pack2<DT>(va[ni][2], va[ni][3]);
        __builtin_amdgcn_raw_buffer_store_b64(o, r_out, oa, 0, MIMO_ST_AUX);
        o.x = pack2<DT>(vb[ni][0], vb[ni][1]); o.y = pack2<DT>(vb[ni][2], vb[ni][3]);
        __builtin_amdgcn_raw_buffer_store_b64(o, r_out, ob, 0, MIMO_ST_AUX);
      }
    }
    // 3: slab statistics (32 rows = rows li of tile a + rows li of tile b) while the stores drain
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int n = N0 + col0 + ni * 16;
      f32x4 mean, m2;
#pragma unroll
      for (int r = 0; r < 4; ++r) mean[r] = row16_sum(va[ni][r] + vb[ni][r]) * (1.0f / 32.0f);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float da = va[ni][r] - mean[r], db = vb[ni][r] - mean[r];
        m2[r] = row16_sum(fmaf(da, da, db * db));
      }
      const unsigned cso = (okc[ni] && li == 0) ? (slab_local * 2u * (unsigned)g.N + (unsigned)n) * 4u : OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mean), r_cs, cso, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m2), r_cs, cso == OOB ? OOB : cso + (unsigned)g.N * 4u, 0, 0);
    }
  }
}

// ---- epilogue variant with a fused LayerNorm second output (the tile holds whole rows: BN == N) ----
// out[m, :] = acc + bias (+ per-image bias) (+ residual)  stored as usual (the fp32 residual stream), and
// ln_out[m, :] = LayerNorm(out[m, :]) * gamma + beta (+ pe[frame]) stored as half: the operand of the next GEMM.
// Replaces a separate LayerNorm launch that re-read the fp32 tensor from HBM.  Row statistics are exact two-pass
// (mean, then centred sum of squares), fp32; a row is spread over the WN waves of a block row, so the partial sums meet
// in LDS (`red`: 2 x BM x WN floats).
template <int DT, int NR, int MT, int BM, int WN>
__device__ __forceinline__ void tile_epilogue_ln(const GemmArgs& g, f32x4 (&acc)[NR][MT], const int64_t M0,
                                                 const int wm, const int wn, const int lg, const int li, float* red) {
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32;
  const bool res_f32 = g.flags & MIMO_EPI_RES_F32;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const unsigned esz_o = out_f32 ? 4u : 2u, esz_r = res_f32 ? 4u : 2u;
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.out + M0 * g.ldo * esz_o, 0, (int)(((rows_valid - 1) * g.ldo + g.N) * esz_o), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
      (char*)const_cast<void*>(g.res) + M0 * g.ldr * esz_r, 0,
      g.res ? (int)(((rows_valid - 1) * g.ldr + g.N) * esz_r) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
  const int64_t nimg = g.img_bias ? (g.M + g.rows_per_img - 1) / g.rows_per_img : 0;
  const __amdgpu_buffer_rsrc_t r_imgb = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.img_bias), 0, g.img_bias ? (int)(((nimg - 1) * g.ldib + g.N) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_ln = __builtin_amdgcn_make_buffer_rsrc(
      (char*)g.ln_out + M0 * g.N * 2, 0, (int)(rows_valid * g.N * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_gam = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.ln_gamma), 0, g.N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_bet = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.ln_beta), 0, g.N * 4, 0x00020000);
  // the whole tile lies in one frame (host: ln_rows_per_frame % BM == 0)
  const int64_t frame = g.ln_pe ? (M0 / g.ln_rows_per_frame) % g.ln_pe_frames : 0;
  const __amdgpu_buffer_rsrc_t r_pe = __builtin_amdgcn_make_buffer_rsrc(
      (void*)const_cast<float*>(g.ln_pe ? g.ln_pe + frame * g.N : nullptr), 0, g.ln_pe ? g.N * 4 : 0, 0x00020000);
  const int row0 = wm * 16 * MT + li;
  const int col0 = wn * 16 * NR + 4 * lg;
  auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MIMO_LD_AUX));
  };
  // ---- 1: the ordinary epilogue; the stored values stay in `acc`.  All loads first, then all stores: interleaved
  // they serialise (vmcnt retires in order, a load behind a store waits for the store's acknowledgement) ----
  {
    f32x4 rr[NR][MT];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const unsigned n = (unsigned)(col0 + ni * 16);
      const f32x4 bv = ld4(r_bias, n * 4u);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const unsigned row = (unsigned)(row0 + mi * 16);
        acc[ni][mi] += bv;
        if (g.img_bias) acc[ni][mi] += ld4(r_imgb, ((unsigned)(M0 + row) / (unsigned)g.rows_per_img) * (unsigned)g.ldib * 4u + n * 4u);
        if (g.res) {
          if (res_f32) {
            rr[ni][mi] = ld4(r_res, (row * (unsigned)g.ldr + n) * 4u);
          } else {
            const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(r_res, (row * (unsigned)g.ldr + n) * 2u, 0, MIMO_LD_AUX);
            rr[ni][mi] = (f32x4){HT<DT>::to_f((uint16_t)(h.x & 0xffffu)), HT<DT>::to_f((uint16_t)(h.x >> 16)),
                                 HT<DT>::to_f((uint16_t)(h.y & 0xffffu)), HT<DT>::to_f((uint16_t)(h.y >> 16))};
          }
        }
      }
    }
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const unsigned n = (unsigned)(col0 + ni * 16);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const unsigned row = (unsigned)(row0 + mi * 16);
        f32x4 v = acc[ni][mi];
        if (g.res) v += rr[ni][mi];
        v *= g.out_scale;
        acc[ni][mi] = v;
        const unsigned ooff = (row * (unsigned)g.ldo + n) * esz_o;
        if (out_f32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, ooff, 0, MIMO_ST_AUX);
        } else {
          u32x2 o;
          o.x = pack2<DT>(v[0], v[1]); o.y = pack2<DT>(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, r_out, ooff, 0, MIMO_ST_AUX);
        }
      }
    }
  }
  // ---- 2: row means.  A lane holds 4 NR values of each of its MT rows; the 4 lane groups lg and the WN waves hold the rest ----
  const float invn = 1.0f / (float)g.N;
  float mean[MT], rstd[MT];
  auto row_reduce = [&](float (&s)[MT], float* redp) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      s[mi] += __shfl_xor(s[mi], 16, 64);
      s[mi] += __shfl_xor(s[mi], 32, 64);
      if (lg == 0) redp[(wm * 16 * MT + mi * 16 + li) * WN + wn] = s[mi];
    }
    // LDS-only rendezvous: a __syncthreads() would also wait (vmcnt) for the epilogue's global stores
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const float* rp = redp + (wm * 16 * MT + mi * 16 + li) * WN;
      float t = rp[0];
#pragma unroll
      for (int w = 1; w < WN; ++w) t += rp[w];  // fixed order
      s[mi] = t;
    }
  };
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    float t = 0.f;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) t += (acc[ni][mi][0] + acc[ni][mi][1]) + (acc[ni][mi][2] + acc[ni][mi][3]);
    mean[mi] = t;
  }
  row_reduce(mean, red);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) mean[mi] *= invn;
  // ---- 3: centred sums of squares ----
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    float t = 0.f;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[ni][mi][r] - mean[mi];
        t = fmaf(d, d, t);
      }
    rstd[mi] = t;
  }
  row_reduce(rstd, red + BM * WN);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) rstd[mi] = rsqrtf(rstd[mi] * invn + g.ln_eps);
  // ---- 4: normalise, affine (+ positional table), store half ----
#pragma unroll
  for (int ni = 0; ni < NR; ++ni) {
    const unsigned n = (unsigned)(col0 + ni * 16);
    const f32x4 gm = ld4(r_gam, n * 4u), bt = ld4(r_bet, n * 4u) + ld4(r_pe, g.ln_pe ? n * 4u : OOB);
#pragma unroll
    for (int mi = 0; mi < MT; mi += 2) {  // 16-byte stores through the lane exchange of tile_epilogue
      u32x2 o[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = fmaf((acc[ni][mi + u][r] - mean[mi + u]) * rstd[mi + u], gm[r], bt[r]);
        o[u].x = pack2<DT>(y[0], y[1]); o[u].y = pack2<DT>(y[2], y[3]);
      }
      const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
      const u32x4 o16 = {sx[0], sy[0], sx[1], sy[1]};
      const unsigned row = (unsigned)(row0 + (mi + (lg & 1)) * 16);
      const unsigned c8 = (unsigned)(wn * 16 * NR + ni * 16 + 4 * (lg & ~1));
      __builtin_amdgcn_raw_buffer_store_b128(o16, r_ln, (row * (unsigned)g.N + c8) * 2u, 0, MIMO_ST_AUX);
    }
  }
}

// MODE 0: dense GEMM; 1: convolution gather; 2: convolution gather through a nearest-neighbour upsampling
// EPI 0: ordinary epilogue (+ GroupNorm column statistics when g.colstats is set); 1: + fused LayerNorm second output
template <int DT, int NR, int MODE, int WM, int WN, int NSTAGE, int MT, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void gemm_kernel(const GemmArgs g) {
  constexpr bool CONV = MODE != 0;
  constexpr int THREADS = 64 * WM * WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NR * WN;
  constexpr int PASS = THREADS / 8;                   // tile rows covered by one DMA pass of the whole block
  constexpr int NAJ = BM / PASS;                      // A passes
  constexpr int NBJ = (BN + PASS - 1) / PASS;         // B passes
  constexpr int BNA = NBJ * PASS > BN ? BN + 8 : BN;  // + 8 dummy rows that absorb the surplus (all-zero) DMAs
  constexpr int LOADS = NAJ + NBJ;                    // DMAs per thread per K-tile
  static_assert(BM % PASS == 0, "A tile must be a whole number of DMA passes");
  constexpr int STAGE = (BM + BNA) * 8;               // 16-byte units per ring slot: [A tile | B tile]
  constexpr int LNRED = EPI == 1 ? (2 * BM * WN) / 4       // 16-byte units of the LayerNorm row-sum exchange
                        : EPI == 2 ? (2 * BM + BN) / 4     // ... of the folded LayerNorm's (mean, rstd) per row and colsum per column
                                   : 0;
  // ONE LDS object (a second __shared__ array would make hipcc drain vmcnt before fragment reads)
  __shared__ __attribute__((aligned(16))) uint4 smem[NSTAGE * STAGE + LNRED];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lg = lane >> 4, li = lane & 15;

  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = (int)(L / (unsigned)g.tiles_n);
  const int tile_n = (int)(L % (unsigned)g.tiles_n);
  const int64_t M0 = (int64_t)tile_m * BM;
  const int N0 = tile_n * BN;
  [[maybe_unused]] u32x4 rsq[EPI == 2 ? ROW_STAT_LOADS : 1], csq;
  if constexpr (EPI == 2) row_stats_issue(g, M0, N0, BM, BN, tid, rsq, csq);   // (all out of range without a_row_stats)

  // ---- staging roles: thread -> (row srow + PASS*j, physical 16-byte chunk tid&7) ----
  // The DMA writes lane-linear (wave-uniform base + lane*16), so the lane that fills physical chunk (tid&7) of
  // a row fetches logical chunk sc = (tid&7) ^ (row&7); (srow + PASS*j) & 7 == srow & 7.
  const int srow = tid >> 3;
  const int sc = (tid & 7) ^ (srow & 7);
  constexpr unsigned OOB = 0xFFFFFFF0u;  // out of every descriptor's range -> the DMA writes zeros

  unsigned a_base[NAJ];  // MODE 0: byte offset of (row, chunk sc); 1: byte offset of tap (0,0); 2: image index
  unsigned a2_base[NAJ];
  int a_iy0[NAJ], a_ix0[NAJ];
  bool a_ok[NAJ];
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int64_t m = M0 + srow + PASS * j;
    a_ok[j] = m < g.M;
    a2_base[j] = 0;
    if (CONV) {
      const int64_t hw = (int64_t)g.Hout * g.Wout;
      const int64_t img = a_ok[j] ? m / hw : 0;
      const int rem = a_ok[j] ? (int)(m - img * hw) : 0;
      const int oy = rem / g.Wout, ox = rem - oy * g.Wout;
      a_iy0[j] = oy * g.stride - g.pad_t;
      a_ix0[j] = ox * g.stride - g.pad_l;
      // arithmetic is modulo 2^32: exact for every in-range final offset
      a_base[j] = MODE == 1 ? (unsigned)((((img * g.Hin + a_iy0[j]) * g.Win + a_ix0[j]) * g.Cin + sc * 8) * 2) : (unsigned)img;
      a2_base[j] = (unsigned)((m * g.Cin2 + sc * 8) * 2);
    } else {
      a_base[j] = (unsigned)((m * g.lda + sc * 8) * 2);
      a_iy0[j] = a_ix0[j] = 0;
    }
  }
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned b_base[NBJ];
  unsigned b_row0[NBJ];  // first tile row (wave-uniform) of this wave's 8-row DMA in pass j
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int r = srow + PASS * j;
    const int n = N0 + r;
    b_base[j] = (r < BN && n < g.N) ? (unsigned)(((int64_t)n * g.ldw + sc * 8) * 2) : OOB;
    const unsigned r0 = 8 * wave_u + PASS * j;
    b_row0[j] = r0 < (unsigned)BN ? r0 : (unsigned)BN;  // surplus passes land in the dummy rows
  }

  // raw buffer descriptor: base, stride 0, num_records (bytes), flags
  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  // One DMA = 64 lanes x 16 B = 8 tile rows; lane l lands at lds_base + 16 l.  Issued through inline asm so that
  // hipcc does NOT track it: with the builtin it drains vmcnt(0) before the first fragment read of every tile
  // (it cannot prove the reads touch another ring slot), which serialises DMA and MFMA.  The kernel counts its
  // DMAs itself (s_waitcnt vmcnt below).  M0 is saved/restored.
  auto dma = [&](const i32x4& r, unsigned off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(r), "s"(lds_base) : "memory");
  };
  const i32x4 rW = make_rsrc(g.W, g.w_bytes);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];

  // split-K: gridDim.y blocks share one output tile, block y reduces K-tiles [kt_lo, kt_hi) into its own fp32
  // partial (the launcher points g.out at the workspace); gridDim.y == 1 is the ordinary case
  const int kt_lo = (int)(((int64_t)g.nkt * blockIdx.y) / gridDim.y);
  const int kt_hi = (int)(((int64_t)g.nkt * (blockIdx.y + 1)) / gridDim.y);

  // MODE 2 loader (the nearest-neighbour gather needs a per-tap source coordinate): issue the LOADS DMAs of K-tile kt
  // into ring slot `slot` (kt >= kt_hi: all-zero DMAs keep the counts uniform)
  auto load_tile_general = [&](int kt, int slot) {
    // offsets are computed on (uniform) branches; the DMAs themselves are issued once, after the join
    unsigned offA[NAJ], kw = 0;
    bool cok = false;
    bool main_tap = true;
    const bool live = kt < kt_hi;
    if (CONV) {
      const int ntap_tiles = g.ks * g.ks * g.chunks1;
      main_tap = kt < ntap_tiles;
      if (main_tap) {
        // K order: channel chunk outer, tap inner (flag 0x80000) -> the 9 shifted re-reads of one input chunk are
        // consecutive K-tiles and hit in L2; or tap outer, chunk inner
        const int ntaps = g.ks * g.ks;
        const bool tap_inner = g.flags & F_TAP_INNER;
        const int tap = tap_inner ? kt % ntaps : kt / g.chunks1;
        const int c0 = (tap_inner ? kt / ntaps : kt - tap * g.chunks1) * BK;
        const int ky = tap / g.ks, kx = tap - ky * g.ks;
        cok = c0 + sc * 8 < g.Cin;
        const int Hv = MODE == 2 ? g.Hup : g.Hin;
        const int Wv = MODE == 2 ? g.Wup : g.Win;
        const unsigned tap_off = (unsigned)((((int64_t)ky * g.Win + kx) * g.Cin + c0) * 2);
#pragma unroll
        for (int j = 0; j < NAJ; ++j) {
          const int vy = a_iy0[j] + ky, vx = a_ix0[j] + kx;
          const bool ok = a_ok[j] & cok & ((unsigned)vy < (unsigned)Hv) & ((unsigned)vx < (unsigned)Wv);
          unsigned off;
          if (MODE == 2) {
            const int sy = min((int)floorf((float)vy * g.sh), g.Hin - 1);
            const int sx = min((int)floorf((float)vx * g.sw), g.Win - 1);
            off = (unsigned)(((((int64_t)a_base[j] * g.Hin + sy) * g.Win + sx) * g.Cin + c0 + sc * 8) * 2);
          } else {
            off = a_base[j] + tap_off;
          }
          offA[j] = ok ? off : OOB;
        }
        kw = (unsigned)(((int64_t)tap * g.Cin + c0) * 2);
      } else {
        const int c0 = (kt - ntap_tiles) * BK;
        cok = live & (c0 + sc * 8 < g.Cin2);
#pragma unroll
        for (int j = 0; j < NAJ; ++j) offA[j] = (a_ok[j] & cok) ? a2_base[j] + (unsigned)(c0 * 2) : OOB;
        kw = (unsigned)(((int64_t)g.ks * g.ks * g.Cin + c0) * 2);
      }
    } else {
      cok = live & (kt * BK + sc * 8 < g.K);
      kw = (unsigned)(kt * BK * 2);
#pragma unroll
      for (int j = 0; j < NAJ; ++j) offA[j] = (a_ok[j] & cok) ? a_base[j] + kw : OOB;
    }
    const i32x4 rsel = make_rsrc(main_tap ? g.A : g.A2, main_tap ? g.a_bytes : g.a2_bytes);
    const unsigned slot_base = smem_base + 16u * (unsigned)(slot * STAGE);
#pragma unroll
    for (int j = 0; j < NAJ; ++j) dma(rsel, offA[j], slot_base + 16u * ((wave_u * 8 + PASS * j) * 8));
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      dma(rW, (cok & (b_base[j] != OOB)) ? b_base[j] + kw : OOB, slot_base + 16u * ((BM + b_row0[j]) * 8));
  };


  // ---- MODE 0 / 1 loader: everything a lane contributes to an address is loop-invariant ----
  // The K-loop of the 256x320 tile has 80 MFMAs per wave per K-tile; a loader that re-derives (tap, chunk) with integer
  // divisions, re-checks the image border per row and rebuilds descriptors per tile spends several hundred scalar/vector
  // instructions per K-tile and the kernel becomes issue-bound.  Here a lane keeps, per A pass, the byte offset of tap
  // (0,0) of its row and a 9-bit mask of the taps that fall inside the image; the K-tile contributes a scalar tap offset
  // (one v_add) and two scalar `soffset`s (channel chunk, weight column); the B offsets never change.  Invalid lanes use
  // OOBA: every descriptor is < 2 GiB (checked on the host), so OOBA + soffset is out of range whether or not the
  // hardware adds soffset before the range check.
  constexpr unsigned OOBA = 0x80000000u;
  unsigned fa_mask[NAJ], fb_base[NBJ];
  if constexpr (MODE != 2) {
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      unsigned mk = 0;
      if (CONV) {
        for (int ky = 0; ky < g.ks; ++ky)
          for (int kx = 0; kx < g.ks; ++kx)
            if (a_ok[j] && (unsigned)(a_iy0[j] + ky) < (unsigned)g.Hin && (unsigned)(a_ix0[j] + kx) < (unsigned)g.Win)
              mk |= 1u << (ky * g.ks + kx);
        if (!a_ok[j]) a2_base[j] = OOBA;
      } else if (!a_ok[j]) {
        a_base[j] = OOBA;
      }
      fa_mask[j] = mk;
    }
#pragma unroll
    for (int j = 0; j < NBJ; ++j) fb_base[j] = b_base[j] == OOB ? OOBA : b_base[j];
  }
  const i32x4 rA1 = make_rsrc(g.A, g.a_bytes);
  const i32x4 rA2 = make_rsrc(g.A2, g.a2_bytes);
  // lanes whose 16-byte chunk exists in the LAST (ragged) channel chunk of a segment
  const bool tail1_ok = CONV ? (g.chunks1 - 1) * BK + sc * 8 < g.Cin : (g.nkt - 1) * BK + sc * 8 < g.K;
  const bool tail2_ok = CONV ? (g.chunks2 - 1) * BK + sc * 8 < g.Cin2 : true;
  const bool ragged1 = CONV ? (g.Cin & (BK - 1)) != 0 : (g.K & (BK - 1)) != 0;
  const bool ragged2 = CONV && (g.Cin2 & (BK - 1)) != 0;
  // scalar cursor: the K-tile the next load_next() fetches
  int ld_kt = kt_lo, ld_tap = 0, ld_chunk = 0;  // conv: ld_tap == ntaps marks the shortcut segment
  const int ntaps = CONV ? g.ks * g.ks : 1;
  const bool tap_inner = g.flags & F_TAP_INNER;
  if (CONV && MODE != 2) {
    const int ntt = ntaps * g.chunks1;
    if (kt_lo >= ntt) { ld_tap = ntaps; ld_chunk = kt_lo - ntt; }
    else if (tap_inner) { ld_chunk = kt_lo / ntaps; ld_tap = kt_lo - ld_chunk * ntaps; }
    else { ld_tap = kt_lo / g.chunks1; ld_chunk = kt_lo - ld_tap * g.chunks1; }
  }
  auto dma_s = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(lds_base) : "memory", "m0");
  };
  auto load_next = [&](int slot) {
    unsigned offA[NAJ], offB[NBJ], soffA, soffW;
    bool seg2 = false, ragged_now;
#pragma unroll
    for (int j = 0; j < NBJ; ++j) offB[j] = fb_base[j];
    if (CONV) {
      seg2 = ld_tap >= ntaps;
      if (!seg2) {
        const unsigned ky = (unsigned)(ld_tap * 11) >> 5, kx = (unsigned)ld_tap - ky * (unsigned)g.ks;  // ks in {1, 3}
        const unsigned tap_off = (ky * (unsigned)g.Win + kx) * (unsigned)g.Cin * 2u;
        const unsigned bit = 1u << ld_tap;
#pragma unroll
        for (int j = 0; j < NAJ; ++j) offA[j] = (fa_mask[j] & bit) ? a_base[j] + tap_off : OOBA;
        soffA = (unsigned)ld_chunk * (BK * 2u);
        soffW = (unsigned)(ld_tap * g.Cin) * 2u + soffA;
        ragged_now = ragged1 && ld_chunk == g.chunks1 - 1;
      } else {
#pragma unroll
        for (int j = 0; j < NAJ; ++j) offA[j] = a2_base[j];
        soffA = (unsigned)ld_chunk * (BK * 2u);
        soffW = (unsigned)(ntaps * g.Cin) * 2u + soffA;
        ragged_now = ragged2 && ld_chunk == g.chunks2 - 1;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NAJ; ++j) offA[j] = a_base[j];
      soffA = soffW = (unsigned)ld_kt * (BK * 2u);
      ragged_now = ragged1 && ld_kt == g.nkt - 1;
    }
    if (ragged_now) {  // (uniform, rare) the last chunk of a ragged channel count: some lanes have no data
      const bool ok = seg2 ? tail2_ok : tail1_ok;
#pragma unroll
      for (int j = 0; j < NAJ; ++j) offA[j] = ok ? offA[j] : OOBA;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) offB[j] = ok ? offB[j] : OOBA;
    }
    if (ld_kt >= kt_hi) {  // (uniform) past the end: all-zero DMAs keep the per-wave counts uniform
#pragma unroll
      for (int j = 0; j < NAJ; ++j) offA[j] = OOBA;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) offB[j] = OOBA;
    }
    const unsigned slot_base = smem_base + 16u * (unsigned)(slot * STAGE);
    const i32x4 rsel = seg2 ? rA2 : rA1;
#pragma unroll
    for (int j = 0; j < NAJ; ++j) dma_s(rsel, offA[j], soffA, slot_base + 16u * ((wave_u * 8 + PASS * j) * 8));
#pragma unroll
    for (int j = 0; j < NBJ; ++j) dma_s(rW, offB[j], soffW, slot_base + 16u * ((BM + b_row0[j]) * 8));
    // advance the cursor
    ++ld_kt;
    if (CONV) {
      if (seg2) {
        ++ld_chunk;
      } else if (tap_inner) {
        if (++ld_tap == ntaps) {
          ld_tap = 0;
          if (++ld_chunk == g.chunks1) { ld_tap = ntaps; ld_chunk = 0; }
        }
      } else if (++ld_chunk == g.chunks1) {
        ld_chunk = 0;
        ++ld_tap;
      }
    }
  };
  // K-tiles are requested in order (kt_lo, kt_lo + 1, ...): the fast loader ignores `kt` and follows its own cursor
  auto load_tile = [&](int kt, int slot) {
    if constexpr (MODE == 2) load_tile_general(kt, slot);
    else load_next(slot);
  };

  f32x4 acc[NR][MT];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one k32 half (s = 0 | 1) of the K-tile in ring slot S
  auto compute = [&](auto slot_c, int s) {
    constexpr int S = decltype(slot_c)::value;
    const uint4* sa = &smem[S * STAGE];
    const uint4* sb = &smem[S * STAGE + BM * 8];
    const int ch = (4 * s + lg) ^ (li & 7);
    uint4 fa[MT], fb[NR];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) fa[mi] = sa[(wm * 16 * MT + mi * 16 + li) * 8 + ch];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) fb[ni] = sb[(wn * 16 * NR + ni * 16 + li) * 8 + ch];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        // swapped: D[row = n-in-tile = 4*lg + r][col = m-in-tile = li]
        acc[ni][mi] = HT<DT>::mfma16(fb[ni], fa[mi], acc[ni][mi]);
  };
  // Waves that share a SIMD (w, w+4, ...) issue their DMAs at different points of the K-tile, so that a
  // SIMD's matrix pipe is not left idle while all of its waves sit in the (slow-to-issue) DMA instructions.
  const bool late = (g.flags & F_STAGGER) && ((wave_u >> 2) & 1);  // wave_u: the branch must be uniform for the compiler too

  // One K-tile: wait until everything but the newest NSTAGE-2 tiles of THIS wave has landed, barrier (all
  // waves' parts landed AND every wave is done reading the slot about to be recycled), refill that slot with
  // tile kt + NSTAGE - 1, then run the MFMAs of tile kt while the DMAs fly.
  unsigned tr = 0;
  auto step = [&](auto slot_c, int kt) {
    constexpr int S = decltype(slot_c)::value;
    MIMO_TRACE(g, tr, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LOADS) : "memory");
    MIMO_TRACE(g, tr, 7);
    __syncthreads();
    MIMO_TRACE(g, tr, 3);
    const bool dma_on = !MIMO_ABLATE(g, F_ABL_NO_DMA), mfma_on = !MIMO_ABLATE(g, F_ABL_NO_MFMA);
    if (dma_on && !late) load_tile(kt + NSTAGE - 1, (S + NSTAGE - 1) % NSTAGE);
    if (mfma_on) compute(slot_c, 0);
    if (dma_on && late) load_tile(kt + NSTAGE - 1, (S + NSTAGE - 1) % NSTAGE);
    if (mfma_on) compute(slot_c, 1);
    MIMO_TRACE(g, tr, 4);
  };

  MIMO_TRACE_REAL(g, tr, 0xfe);
  MIMO_TRACE(g, tr, 1);
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i) load_tile(kt_lo + i, i);
  if constexpr (EPI == 2) {   // (the K loop's barriers order these LDS writes before the epilogue's reads)
    if (g.a_row_stats) {
      float* mr = reinterpret_cast<float*>(&smem[NSTAGE * STAGE]);
      row_stats_finish(g, BM, BN, tid, rsq, csq, mr, mr + 2 * BM);
    }
  }
  for (int kt = kt_lo; kt < kt_hi; kt += NSTAGE) {
    step(IC<0>{}, kt);
    if (kt + 1 < kt_hi) step(IC<1 % NSTAGE>{}, kt + 1);
    if (NSTAGE > 2 && kt + 2 < kt_hi) step(IC<2 % NSTAGE>{}, kt + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
  MIMO_TRACE(g, tr, 5);

  if constexpr (EPI == 1) {
    tile_epilogue_ln<DT, NR, MT, BM, WN>(g, acc, M0, wm, wn, lg, li, reinterpret_cast<float*>(&smem[NSTAGE * STAGE]));
  } else if constexpr (EPI == 2) {  // LayerNorm folded into the consumer: the instantiations that carry its two halves
    if (g.a_row_stats) {
      const float* mr = reinterpret_cast<const float*>(&smem[NSTAGE * STAGE]);
      tile_row_affine<NR, MT>(acc, wm, wn, lg, li, mr, mr + 2 * BM);
    }
    bool done = false;
    if constexpr (WM * WN <= 8) {
      if (g.row_half) {
        tile_epilogue_rowside<DT, NR, MT, BM>(g, acc, M0, N0, wm, wn, lg, li);
        done = true;
      }
    }
    if (!done) tile_epilogue<DT, NR, MT, BM, (WM * WN <= 8), (WM * WN <= 8)>(g, acc, M0, N0, wm, wn, lg, li, 0u);
  } else {
    bool done = false;
    if constexpr (WM * WN <= 8) {  // (the 16-wave tiles have a 128-register budget: no room for a second epilogue)
      if (g.colstats) {
        tile_epilogue_stats<DT, NR, MT, BM>(g, acc, M0, N0, wm, wn, lg, li);
        done = true;
      }
    }
    if (!done) tile_epilogue<DT, NR, MT, BM, (WM * WN <= 8)>(g, acc, M0, N0, wm, wn, lg, li, blockIdx.y);
  }
#ifdef MIMO_TUNE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MIMO_TRACE(g, tr, 6);
  MIMO_TRACE_REAL(g, tr, 0xff);
}

// Persistent dense GEMM (MODE 0, 2-deep ring): the grid is ONE resident set of blocks; block b walks tiles b, b + grid, ...
// and its DMA cursor runs one K-tile ahead of its MFMA cursor ACROSS tile boundaries, so the first K-tile of the next
// output tile lands under the last MFMAs and the epilogue of the current one (short-K linears: K = 320 is five K-tiles
// per output tile and paid a full L2/HBM round trip plus a block launch per tile).  The loader keeps two VGPRs of state:
// per-tile geometry lives in the (scalar) buffer descriptors, clipped to the tile's valid rows.
template <int DT, int NR, int WM, int WN, int MT, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void gemm_dense_persist_kernel(const GemmArgs g) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr int BM = 16 * MT * WM;
  constexpr int BN = 16 * NR * WN;
  constexpr int PASS = THREADS / 8;
  constexpr int NAJ = BM / PASS;
  constexpr int NBJ = (BN + PASS - 1) / PASS;
  constexpr int BNA = NBJ * PASS > BN ? BN + 8 : BN;
  static_assert(BM % PASS == 0, "A tile must be a whole number of DMA passes");
  constexpr int STAGE = (BM + BNA) * 8;
  constexpr int LNRED = EPI == 1 ? (2 * BM * WN) / 4 : 0;
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * STAGE + LNRED];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lg = lane >> 4, li = lane & 15;
  const int srow = tid >> 3;
  const int sc = (tid & 7) ^ (srow & 7);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ntiles = g.ntiles;
  const int nkt = g.nkt;

  const unsigned a_thr = (unsigned)(((int64_t)srow * g.lda + sc * 8) * 2);  // byte offset inside the tile, pass 0
  const unsigned b_thr = (unsigned)(((int64_t)srow * g.ldw + sc * 8) * 2);
  const unsigned a_pass = (unsigned)(PASS * g.lda * 2), b_pass = (unsigned)(PASS * g.ldw * 2);
  const bool b_tail_ok = srow + PASS * (NBJ - 1) < BN;  // rows of the last (partial) B pass that exist in the tile

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  auto dma = [&](const i32x4& r, unsigned off, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(r), "s"(lds_base) : "memory");
  };
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];

  // loader state: descriptors of the tile the DMA cursor is in (all scalar)
  i32x4 rA, rW;
  auto set_tile = [&](unsigned v) {
    const bool valid = v < ntiles;
    const unsigned L = valid ? xcd_remap(v, ntiles) : 0u;
    const int64_t M0 = (int64_t)(L / (unsigned)g.tiles_n) * BM;
    const int N0 = (int)(L % (unsigned)g.tiles_n) * BN;
    const int64_t ra = valid ? ((g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM) : 0;
    const int rb = valid ? ((g.N - N0) < BN ? (g.N - N0) : BN) : 0;
    rA = make_rsrc(g.A + M0 * g.lda, ra > 0 ? (unsigned)(((ra - 1) * g.lda + g.K) * 2) : 0u);
    rW = make_rsrc(g.W + (int64_t)N0 * g.ldw, rb > 0 ? (unsigned)(((int64_t)(rb - 1) * g.ldw + g.K) * 2) : 0u);
  };
  unsigned ld_v = blockIdx.x;
  int ld_kt = 0;
  set_tile(ld_v);
  // lane offsets inside the tile never change (the K-tile goes into the scalar soffset; rows beyond the tile are clipped
  // by the descriptors, whose extents are < 2 GiB so that OOBA + soffset stays out of range)
  constexpr unsigned OOBA = 0x80000000u;
  unsigned pa_off[NAJ], pb_off[NBJ];
#pragma unroll
  for (int j = 0; j < NAJ; ++j) pa_off[j] = a_thr + j * a_pass;
#pragma unroll
  for (int j = 0; j < NBJ; ++j) pb_off[j] = (j < NBJ - 1 || b_tail_ok) ? b_thr + j * b_pass : OOBA;
  const bool ragged = (g.K & (BK - 1)) != 0;
  const bool tail_ok = (nkt - 1) * BK + sc * 8 < g.K;
  auto dma_s = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(lds_base) : "memory", "m0");
  };
  auto issue = [&](int slot) {
    const unsigned kw = (unsigned)(ld_kt * BK * 2);
    unsigned offA[NAJ], offB[NBJ];
#pragma unroll
    for (int j = 0; j < NAJ; ++j) offA[j] = pa_off[j];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) offB[j] = pb_off[j];
    if (ragged && ld_kt == nkt - 1) {  // (uniform) only the last K-tile of a ragged K has lanes without data
#pragma unroll
      for (int j = 0; j < NAJ; ++j) offA[j] = tail_ok ? offA[j] : OOBA;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) offB[j] = tail_ok ? offB[j] : OOBA;
    }
    const unsigned slot_base = smem_base + 16u * (unsigned)(slot * STAGE);
#pragma unroll
    for (int j = 0; j < NAJ; ++j) dma_s(rA, offA[j], kw, slot_base + 16u * ((wave_u * 8 + PASS * j) * 8));
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const unsigned r0 = 8 * wave_u + PASS * j;
      dma_s(rW, offB[j], kw, slot_base + 16u * ((BM + (r0 < (unsigned)BN ? r0 : (unsigned)BN)) * 8));
    }
    if (++ld_kt == nkt) {
      ld_kt = 0;
      ld_v += gridDim.x;
      set_tile(ld_v);
    }
  };
  const bool late = (g.flags & F_STAGGER) && ((wave_u >> 2) & 1);

  unsigned tr = 0;
  MIMO_TRACE_REAL(g, tr, 0xfe);
  MIMO_TRACE(g, tr, 1);
  issue(0);
  int slot = 0;
  for (unsigned cv = blockIdx.x; cv < ntiles; cv += gridDim.x) {
    f32x4 acc[NR][MT];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
      // vmcnt(0): with a 2-deep ring every wait is a full drain anyway; it also retires the previous epilogue's stores
      // (waiting at the END of the K-tile instead, so that the stores retire under the next tile's first MFMAs,
      // measured 3-8 % slower)
      MIMO_TRACE(g, tr, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      MIMO_TRACE(g, tr, 7);
      __syncthreads();
      MIMO_TRACE(g, tr, 3);
      const uint4* sa = &smem[slot * STAGE];
      const uint4* sb = sa + BM * 8;
      if (!late) issue(slot ^ 1);
#pragma unroll
      for (int sh = 0; sh < 2; ++sh) {
        if (sh == 1 && late) issue(slot ^ 1);
        const int ch = (4 * sh + lg) ^ (li & 7);
        uint4 fa[MT], fb[NR];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) fa[mi] = sa[(wm * 16 * MT + mi * 16 + li) * 8 + ch];
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) fb[ni] = sb[(wn * 16 * NR + ni * 16 + li) * 8 + ch];
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = HT<DT>::mfma16(fb[ni], fa[mi], acc[ni][mi]);
      }
      slot ^= 1;
      MIMO_TRACE(g, tr, 4);
    }
    MIMO_TRACE(g, tr, 5);
    const unsigned Lc = xcd_remap(cv, ntiles);
    const int64_t M0 = (int64_t)(Lc / (unsigned)g.tiles_n) * BM;
    const int N0 = (int)(Lc % (unsigned)g.tiles_n) * BN;
    int lg_ = lg, li_ = li;
    asm volatile("" : "+v"(lg_), "+v"(li_));  // keeps the epilogue's lane-invariant address math inside the tile loop
    if constexpr (EPI == 1) {
      tile_epilogue_ln<DT, NR, MT, BM, WN>(g, acc, M0, wm, wn, lg_, li_, reinterpret_cast<float*>(&smem[2 * STAGE]));
    } else {
      tile_epilogue<DT, NR, MT, BM, (WM * WN <= 8)>(g, acc, M0, N0, wm, wn, lg_, li_, 0u);
    }
    MIMO_TRACE(g, tr, 6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
  MIMO_TRACE_REAL(g, tr, 0xff);
}

constexpr int G8_GM = 4;   // gemm8_kernel: row panels per tile-order group (tune build: MIMO_G8_GM; 1 = plain row-major order)

// Dense GEMM, 256 x 256 x 64 tile, EIGHT phases per pair of K-tiles with the two wave groups half a phase apart
// (cdna_hip_programming.md, "The 256^2 8-phase template"): 8 waves = 2 (M) x 4 (N), each 128 x 64 of the tile (128 accumulator
// registers).  A K-tile is four phases, one 64 x 32 quadrant of the wave's output each (16 MFMAs):
//   LOAD segment: ds_read the quadrant's operand sub-tile that is not in registers yet (8 A | 4 B fragments), issue the
//                 LDS-DMAs of ONE half-tile (128 rows x 64) of a K-tile ahead, lgkmcnt(0), barrier
//   MFMA segment: 16 MFMAs at raised priority, barrier
// Waves 0-3 (group 0) and 4-7 (group 1, its SIMD partners) run ONE barrier apart: while a group multiplies, the other one
// reads LDS and issues DMAs, so a SIMD's matrix pipe sees MFMAs from one wave at a time and the partner's load segment
// hides under them.  LDS = 8 half-tile slots of 16 KB (K-tile parity x {A, B} x half); DMAs are counted (vmcnt(4) once
// per K-tile, never 0 in the loop).  Slot hand-off (reads of a phase are complete before its barrier):
//   B halves of K-tile t+1 are staged in phases 0, 1 of K-tile t (their slots were last read in phase 3 of t-1),
//   A halves of K-tile t+2 in phase 3 of t (A sub-tiles are read in phases 0 and 2 only), followed by the wait that
//   retires everything K-tile t+1 needs; the first reads of t+1 come two barriers later in either group.
template <int DT, int EPI = 0>   // EPI 2: + the two halves of a LayerNorm folded into the consumer (tile_row_affine / tile_epilogue_rowside)
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const GemmArgs g) {
  constexpr int NR = 4, MT = 8, BM = 256, BN = 256, HTB = 16384;  // HTB: bytes of a half-tile
  // (EPI 2: + (mean, rstd) of the 256 rows and the colsum of the 256 columns of a folded LayerNorm)
  __shared__ __attribute__((aligned(16))) uint4 smem[8 * HTB / 16 + (EPI == 2 ? (2 * BM + BN) / 4 : 0)];
  char* const lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = (int)(wave_u >> 2), wc = (int)(wave_u & 3u);
  const int lg = lane >> 4, li = lane & 15;
  // Tile order.  The 32 tiles an XCD runs at a time walk K in step, so its L2 serves every operand panel they SHARE once: with
  // the plain row-major order (tile_n fastest) a window is ONE A panel x 32 W tiles — 33 panel streams for 32 tiles, and a wide
  // GEMM (N = 10240: 40 column tiles) re-fetches the whole weight for every A panel (measured: 1.33 GB fetched per launch for
  // 58 MB of operands, profiles/r6_pmc_traffic_by_kernel_before_grouping.txt).  Groups of G8_GM row panels walked row-fastest
  // make the window G8_GM x (32 / G8_GM): 12 streams for 32 tiles.
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  unsigned tile_m, tile_n;
  {
    const unsigned tn = (unsigned)g.tiles_n, tm = gridDim.x / tn;          // (grid = tiles_m x tiles_n exactly)
    const unsigned gm = g.ntiles;
    if (gm <= 1u || tn <= 8u) {
      tile_m = L / tn; tile_n = L % tn;
    } else {
      const unsigned per = gm * tn, grp = L / per, first = grp * gm;
      const unsigned rows = tm - first < gm ? tm - first : gm;              // the last group may be short
      const unsigned r = L - grp * per;
      tile_m = first + r % rows; tile_n = r / rows;
    }
  }
  const int64_t M0 = (int64_t)tile_m * BM;
  const int N0 = (int)tile_n * BN;
  const int nkt = g.nkt;  // even (host)
  [[maybe_unused]] u32x4 rsq[EPI == 2 ? ROW_STAT_LOADS : 1], csq;
  if constexpr (EPI == 2) row_stats_issue(g, M0, N0, BM, BN, tid, rsq, csq);   // (all out of range without a_row_stats)

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  // one descriptor per half-tile row range, clipped to the rows that exist (rows beyond M / N are zero-filled)
  i32x4 rA[2], rB[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int64_t ra = g.M - (M0 + 128 * h), rb = (int64_t)g.N - (N0 + 128 * h);
    const int64_t va = ra < 0 ? 0 : (ra > 128 ? 128 : ra), vb = rb < 0 ? 0 : (rb > 128 ? 128 : rb);
    rA[h] = make_rsrc(g.A + (M0 + 128 * h) * g.lda, va > 0 ? (unsigned)(((va - 1) * g.lda + g.K) * 2) : 0u);
    rB[h] = make_rsrc(g.W + ((int64_t)N0 + 128 * h) * g.ldw, vb > 0 ? (unsigned)(((vb - 1) * g.ldw + g.K) * 2) : 0u);
  }
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  // staging role: thread -> row (tid >> 3) (+ 64 for the second DMA), physical chunk tid & 7 = logical chunk ^ (row & 7)
  const unsigned srow = (unsigned)tid >> 3, lc = ((unsigned)tid & 7u) ^ (srow & 7u);
  const unsigned av = srow * (unsigned)(g.lda * 2) + lc * 16u, bv = srow * (unsigned)(g.ldw * 2) + lc * 16u;
  // The second DMA of a half-tile fetches row srow + 64.  Its 64-row stride rides in the LANE offset: only the vector
  // offset is range checked against the clipped descriptor (the scalar offset is not), so a row beyond a ragged half-tile
  // must be out of range in the voffset to be zero-filled instead of read past the end of A / W.
  const unsigned av2 = av + (unsigned)(g.lda * 128), bv2 = bv + (unsigned)(g.ldw * 128);
  auto dma = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_base))
                 : "memory", "m0");
  };
  // half-tile (operand o, half h) of K-tile kt into parity d: two DMAs per thread
  auto stage = [&](int o, int h, int kt, int d) {
    i32x4 r = o ? rB[h] : rA[h];
    r.z = kt < nkt ? r.z : 0;  // past the end: zero fill keeps the DMA counts uniform
    const unsigned soff = kt < nkt ? (unsigned)kt * 128u : 0u;
    const unsigned base = smem_base + (unsigned)(((d * 2 + o) * 2 + h) * HTB) + wave_u * 1024u;
    dma(r, o ? bv : av, soff, base);
    dma(r, o ? bv2 : av2, soff, base + 8192u);
  };
  // fragment bases: row li of a 16-row tile, chunk (4 kh + lg) ^ (li & 7); the wave's half-tile and column offset folded in
  unsigned ab[2][2], bb[2][2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const unsigned ch = (unsigned)((4 * kh + lg) ^ (li & 7));
      ab[d][kh] = (unsigned)(((d * 2 + 0) * 2 + wr) * HTB) + (unsigned)li * 128u + ch * 16u;
      bb[d][kh] = (unsigned)(((d * 2 + 1) * 2 + (wc >> 1)) * HTB) + (unsigned)(((wc & 1) * 64 + li) * 128) + ch * 16u;
    }

  f32x4 acc[NR][MT];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 fa[4][2], fb[2][2];
  auto read_a = [&](auto d_c, int mq) {
    constexpr int D = decltype(d_c)::value;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi][kh] = *reinterpret_cast<const uint4*>(lds + ab[D][kh] + (mq * 64 + mi * 16) * 128);
  };
  auto read_b = [&](auto d_c, int nq) {
    constexpr int D = decltype(d_c)::value;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[ni][kh] = *reinterpret_cast<const uint4*>(lds + bb[D][kh] + (nq * 32 + ni * 16) * 128);
  };
  auto mfmas = [&](auto mq_c, auto nq_c) {
    constexpr int MQ = decltype(mq_c)::value, NQ = decltype(nq_c)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[NQ * 2 + ni][MQ * 4 + mi] = HT<DT>::mfma16(fb[ni][kh], fa[mi][kh], acc[NQ * 2 + ni][MQ * 4 + mi]);
    __builtin_amdgcn_s_setprio(0);
  };
  // the end of a LOAD segment: this wave's fragment reads are complete BEFORE the barrier (a slot may be re-staged by the
  // other group right after it), and nothing is scheduled across
  auto seg_end = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
  };

  // ---- prologue: K-tile 0 complete, the A halves of K-tile 1 in flight ----
  stage(0, 0, 0, 0); stage(0, 1, 0, 0); stage(1, 0, 0, 0); stage(1, 1, 0, 0);
  if constexpr (EPI == 2) {
    // the statistics were requested before K-tile 0: the compiler's wait for them is the wait for K-tile 0 the first phase
    // needs anyway; the K loop's barriers order these LDS writes before the epilogue's reads
    if (g.a_row_stats) {
      float* mr = reinterpret_cast<float*>(lds + 8 * HTB);
      row_stats_finish(g, BM, BN, tid, rsq, csq, mr, mr + 2 * BM);
    }
  }
  stage(0, 0, 1, 1); stage(0, 1, 1, 1);
  asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
  if (wr == 1) asm volatile("s_barrier" ::: "memory");  // group 1 runs one barrier behind group 0 from here on

  auto ktile = [&](auto d_c, int t) {
    constexpr int D = decltype(d_c)::value;
    // phase 0: quadrant (m0, n0)
    read_b(d_c, 0);
    read_a(d_c, 0);
    stage(1, 0, t + 1, D ^ 1);
    seg_end();
    mfmas(IC<0>{}, IC<0>{});
    mfma_end();
    // phase 1: (m0, n1)
    read_b(d_c, 1);
    stage(1, 1, t + 1, D ^ 1);
    seg_end();
    mfmas(IC<0>{}, IC<1>{});
    mfma_end();
    // phase 2: (m1, n1)
    read_a(d_c, 1);
    seg_end();
    mfmas(IC<1>{}, IC<1>{});
    mfma_end();
    // phase 3: (m1, n0); the A slots of this parity are free (last read in phase 2): K-tile t + 2 moves in
    read_b(d_c, 0);
    stage(0, 0, t + 2, D);
    stage(0, 1, t + 2, D);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything K-tile t + 1 needs has landed (this wave's part)
    seg_end();
    mfmas(IC<1>{}, IC<0>{});
    mfma_end();
  };
  for (int t = 0; t < nkt; t += 2) {
    ktile(IC<0>{}, t);
    ktile(IC<1>{}, t + 1);
  }
  if (wr == 0) asm volatile("s_barrier" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-fill DMAs must not outlive the block's LDS
  if constexpr (EPI == 2) {
    if (g.a_row_stats) {
      const float* mr = reinterpret_cast<const float*>(lds + 8 * HTB);
      tile_row_affine<NR, MT>(acc, wr, wc, lg, li, mr, mr + 2 * BM);
    }
    if (g.row_half) tile_epilogue_rowside<DT, NR, MT, BM>(g, acc, M0, N0, wr, wc, lg, li);
    else tile_epilogue<DT, NR, MT, BM, true, true>(g, acc, M0, N0, wr, wc, lg, li, 0u);
  } else {
    tile_epilogue<DT, NR, MT, BM, true>(g, acc, M0, N0, wr, wc, lg, li, 0u);
  }
}

// Dense GEMM whose tile holds WHOLE rows at N = 640 (level 1 of the UNets), so that the LayerNorm that follows the projection
// (attention.py:329-360 norm1 / norm3 after proj_in / to_out; motion_module.py:230-258 norms + positional table) rides in its
// epilogue (tile_epilogue_ln) and the separate LayerNorm pass — a full read of the fp32 tensor — disappears (round 5).
// A 64 x 640 tile with the 64-deep K-tile of gemm_kernel would need 2 x 90 KB of LDS; here the K-tile is ONE MFMA k-step (32):
// a 16-row x 64-byte piece is 1 KB = one LDS-DMA instruction, stored in MFMA fragment order (the lane that will read a 16-byte
// chunk is the lane that fetches it): fragment reads are contiguous, conflict-free, and nothing is swizzled.  8 waves side by side (64 rows x 80 columns each: MT = 4, NR = 5, 80 accumulator
// registers), 3-deep ring of 44 KB slots, one barrier per K-tile, counted vmcnt (5 W pieces per wave and K-tile + 1 A piece for
// waves 0-3).  The launch is bound by its epilogue traffic (residual in, fp32 + half out), not by the matrix pipe.
template <int DT>
__global__ __launch_bounds__(512, 2) void gemm_ln640_kernel(const GemmArgs g) {
  constexpr int NR = 5, MT = 4, BM = 64, BN = 640, KT = 32, NST = 3;
  constexpr int A_BYTES = BM * KT * 2, W_BYTES = BN * KT * 2, SLOT = A_BYTES + W_BYTES;   // 4 KB + 40 KB
  __shared__ __attribute__((aligned(16))) uint4 smem[(NST * SLOT + 2 * BM * 8 * 4) / 16];  // ONE LDS object: ring | LayerNorm row sums
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = (int)wave_u;
  const int lg = lane >> 4, li = lane & 15;
  const int64_t M0 = (int64_t)blockIdx.x * BM;
  const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
  const int nkt = g.K / KT;

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const i32x4 rA = make_rsrc(g.A + M0 * g.lda, (unsigned)(((rows_valid - 1) * g.lda + g.K) * 2));   // rows beyond M: zero fill
  const i32x4 rW = make_rsrc(g.W, g.w_bytes);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  // piece = 16 tile rows x 64 B = 1 KB.  The DMA writes lane-linear, and the MFMA fragment of lane (li, lg) is row li, chunk lg:
  // lane l = 16 lg + li FETCHES row l & 15, 16-byte chunk l >> 4, so that a piece in LDS is already in fragment order
  // (fragment reads are one contiguous, conflict-free KB)
  const unsigned a_lane = (unsigned)((lane & 15) * g.lda * 2 + (lane >> 4) * 16);
  const unsigned w_lane = (unsigned)((lane & 15) * g.ldw * 2 + (lane >> 4) * 16);
  const bool has_a = wave_u < 4u;   // waves 0-3 also move one of the four A pieces
  auto dma = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory", "m0");
  };
  constexpr unsigned OOBA = 0x80000000u;
  auto issue = [&](int kt) {   // K-tile kt into slot kt % NST (past the end: zero-fill DMAs keep the counts uniform)
    const bool live = kt < nkt;
    const unsigned slot = smem_base + (unsigned)((kt % NST) * SLOT);
    const unsigned kw = (unsigned)(kt * KT * 2);
    if (has_a) dma(rA, live ? a_lane + wave_u * (unsigned)(16 * g.lda * 2) : OOBA, kw, slot + wave_u * 1024u);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const unsigned p = wave_u + 8u * (unsigned)i;   // W piece: rows 16 p .. 16 p + 15
      dma(rW, live ? w_lane + p * (unsigned)(16 * g.ldw * 2) : OOBA, kw, slot + (unsigned)A_BYTES + p * 1024u);
    }
  };
  auto wait_all_but_newest = [&]() {
    if (has_a) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  };

  f32x4 acc[NR][MT];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0);
  issue(1);
  for (int kt = 0; kt < nkt; ++kt) {
    wait_all_but_newest();   // tile kt has landed (this wave's part); tile kt + 1 may stay in flight
    __syncthreads();         // ... every wave's part, and every wave is done reading slot (kt + 2) % 3 = (kt - 1) % 3
    issue(kt + 2);
    const uint4* sa = &smem[((kt % NST) * SLOT) / 16];
    const uint4* sb = sa + A_BYTES / 16;
    uint4 fa[MT], fb[NR];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) fa[mi] = sa[mi * 64 + lane];                    // piece mi in fragment order (row li, chunk lg at 16 lane)
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) fb[ni] = sb[(wn * NR + ni) * 64 + lane];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = HT<DT>::mfma16(fb[ni], fa[mi], acc[ni][mi]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing zero-fill DMAs must not outlive the block's LDS
  __syncthreads();                                   // (the row-sum exchange reuses no ring bytes, but keep the epilogue behind every MFMA read)
  tile_epilogue_ln<DT, NR, MT, BM, 8>(g, acc, M0, 0, wn, lg, li, reinterpret_cast<float*>(&smem[(NST * SLOT) / 16]));
}

// Split-K reduction + the full epilogue: out = epi(sum_s partial[s]); one thread per 4 consecutive columns.
template <int DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs g, int splits) {
  const int n4 = g.N >> 2;
  const int64_t total = g.M * n4;
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32, res_f32 = g.flags & MIMO_EPI_RES_F32, do_silu = g.flags & MIMO_EPI_SILU;
  const float* ws = (const float*)g.ws;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n4;
    const int n = (int)(i - m * n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(ws + m * g.N + n);
    for (int s = 1; s < splits; ++s) {  // fixed order: deterministic
      const float4 p = *reinterpret_cast<const float4*>(ws + ((int64_t)s * g.M + m) * g.N + n);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (g.bias) {
      const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (g.img_bias) {
      const float4 b = *reinterpret_cast<const float4*>(g.img_bias + (m / g.rows_per_img) * g.ldib + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (do_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    if (g.res) {
      if (res_f32) {
        const float4 r = *reinterpret_cast<const float4*>((const float*)g.res + m * g.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      } else {
        const uint2 r = *reinterpret_cast<const uint2*>((const uint16_t*)g.res + m * g.ldr + n);
        v.x += HT<DT>::to_f((uint16_t)(r.x & 0xffffu)); v.y += HT<DT>::to_f((uint16_t)(r.x >> 16));
        v.z += HT<DT>::to_f((uint16_t)(r.y & 0xffffu)); v.w += HT<DT>::to_f((uint16_t)(r.y >> 16));
      }
    }
    v.x *= g.out_scale; v.y *= g.out_scale; v.z *= g.out_scale; v.w *= g.out_scale;
    if (out_f32) {
      *reinterpret_cast<float4*>((float*)g.out + m * g.ldo + n) = v;
    } else {
      uint2 o;
      o.x = pack2<DT>(v.x, v.y);
      o.y = pack2<DT>(v.z, v.w);
      *reinterpret_cast<uint2*>((uint16_t*)g.out + m * g.ldo + n) = o;
    }
  }
}

// Tuning knobs.  In the shipped build they are compile-time constants (tune_env folds to its default: the library
// reads no environment and keeps no mutable state).  Only the -DMIMO_TUNE build (libmimo_hip_tune.so, used by
// tools/microbench.py for interleaved A/B timing) reads them from the environment, at every launch:
//   MIMO_GEMM_CFG=1|2|3|4   force tile configuration S|L|XL|XL8
//   MIMO_GEMM_STAGGER=0     all waves issue their DMAs right after the barrier (default 1: staggered)
//   MIMO_CONV_TAP_INNER=0   convolution K order tap-outer / channel-chunk-inner (default 1: tap inner)
//   MIMO_GEMM_BM=256|192|128  force the XL8 tile height (default: picked per shape by wave quantisation)
//   MIMO_GEMM_PERSIST=0     dense GEMMs launch one block per output tile (default 1: persistent blocks, cross-tile prefetch)
//   MIMO_GEMM_SPLITK=0      never split K (default 1; per call: MIMO_EPI_NO_SPLITK)
//   MIMO_GEMM_ABLATE=1|2|3  timing experiments: skip DMA | skip MFMA | skip GELU (results are wrong)
struct Tuning {
  int cfg, ablate, stagger, tap_inner, splitk, bm, persist;
};
inline Tuning tuning() {
  return Tuning{tune_env("MIMO_GEMM_CFG", 0), tune_env("MIMO_GEMM_ABLATE", 0), tune_env("MIMO_GEMM_STAGGER", 1),
                tune_env("MIMO_CONV_TAP_INNER", 1), tune_env("MIMO_GEMM_SPLITK", 1), tune_env("MIMO_GEMM_BM", 0),
                tune_env("MIMO_GEMM_PERSIST", 1)};
}

// CU count of the current device: an immutable fact of the hardware, cached after the first query
inline int cus_() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

#ifdef MIMO_TUNE
inline unsigned long long* trace_buf() {
  static unsigned long long* buf = [] {
    void* p = nullptr;
    if (hipMalloc(&p, 4096 * sizeof(unsigned long long)) != hipSuccess) return (unsigned long long*)nullptr;
    (void)hipMemset(p, 0, 4096 * sizeof(unsigned long long));
    return (unsigned long long*)p;
  }();
  return buf;
}
extern "C" unsigned long long* mimo_tune_trace_buf() { return trace_buf(); }   // (for the other translation units)
// copies the trace of the most recent traced launch to `dst` (n entries <= 4096) and clears the device buffer
extern "C" int mimo_tune_trace(unsigned long long* dst, int n) {
  if (!trace_buf() || n > 4096) return MIMO_EINVAL;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(dst, trace_buf(), (size_t)n * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemset(trace_buf(), 0, 4096 * 8);
  return (int)e;
}
#endif

// one-block-per-tile launch of gemm_kernel; dense launches that carry a folded LayerNorm half take the EPI = 2 instantiation
#define MIMO_LAUNCH_GK(WM_, WN_, NS_, MT_, nwg_, thr_)                                                                    \
  do {                                                                                                                    \
    if constexpr (MODE == 0) {                                                                                            \
      if (folded) {                                                                                                       \
        hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, WM_, WN_, NS_, MT_, 2>), dim3((unsigned)(nwg_)), dim3(thr_), 0, st, g); \
        break;                                                                                                            \
      }                                                                                                                   \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, WM_, WN_, NS_, MT_>), dim3((unsigned)(nwg_)), dim3(thr_), 0, st, g);      \
  } while (0)

constexpr int GEMM_BM64_DEFAULT = 1;   // (tune build: MIMO_GEMM_BM64)

template <int DT, int MODE, int NR>
int launch_nr(GemmArgs& g, hipStream_t st) {
#ifdef MIMO_TUNE
  g.dbg = tune_env("MIMO_GEMM_TRACE", 0) ? trace_buf() : nullptr;
#endif
  // Tile configurations (WM x WN waves, ring depth):
  //   S  2x2, 2 stages: 128 x 32NR tile, 2 blocks/CU          — small / skinny problems
  //   L  4x2, 3 stages: 256 x 32NR tile, 1 block/CU
  //   XL 4x4, 2 stages: 256 x 64NR tile (N = 320 in ONE tile), 16 waves, 1 block/CU — halves the global->LDS
  //      bytes per MAC; the L2->LDS path (~18 TB/s measured) and not the MFMAs bounds the S tile
  //   XL8 2x4, 2 stages: the XL tile on 8 waves of 128 x 16NR (256-register budget) — every NR = 5 XL problem
  const Tuning tn = tuning();
  const int forced = tn.cfg, ablate = tn.ablate;
  if (ablate == 1) g.flags |= F_ABL_NO_DMA;
  if (ablate == 2) g.flags |= F_ABL_NO_MFMA;
  if (ablate == 3) g.flags |= F_ABL_NO_GELU;
  const int64_t m256 = (g.M + 255) / 256;
  const int tn_s = (g.N + 32 * NR - 1) / (32 * NR), tn_xl = (g.N + 64 * NR - 1) / (64 * NR);
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
  int cfg = forced;
  if (cfg == 0) {
    if (g.N <= 32 * NR) cfg = (m256 * tn_s >= 160) ? 2 : 1;  // one S-width tile covers N: a 64NR-wide XL tile would idle
    else cfg = (m256 * tn_xl >= 160) ? 3 : 1;                 // measured crossover (tools/microbench.py)
  }
  if (tn.stagger) g.flags |= F_STAGGER;
  if (tn.tap_inner && MODE != 0) g.flags |= F_TAP_INNER;
  if (cfg == 3 && NR == 5) cfg = 4;  // 16 waves x 128 registers cannot hold a 64 x 80 accumulator tile plus the epilogue
  if (cfg == 3 && g.colstats) cfg = 4;  // the 16-wave tile has no statistics epilogue
  const bool folded = g.row_half || g.a_row_stats;   // (plain one-block-per-tile launches only)
  if (cfg == 3 && g.row_half) cfg = 4;  // the 16-wave tile has no row-statistics epilogue either
  if (cfg == 2 && folded) cfg = 1;      // the 3-stage ring of the L tile leaves no LDS for the (mean, rstd, colsum) region
  const bool allow_splitk = tn.splitk && !(g.flags & MIMO_EPI_NO_SPLITK) && !g.colstats && !g.ln_out && !folded;
  // fused LayerNorm output: the tile must hold whole rows (N == 320 = the NR = 5 XL8 width), dense only
  if (g.ln_out) {
    if constexpr (MODE == 0 && NR == 5) {
      if (g.N == 640 && !geglu && !g.colstats && (g.K % 32) == 0 && g.M < 0x7fffffffLL * 64) {   // whole rows at N = 640: 64-row tiles
        hipLaunchKernelGGL((gemm_ln640_kernel<DT>), dim3((unsigned)((g.M + 63) / 64)), dim3(512), 0, st, g);
        MIMO_LAUNCH_CHECK();
        return MIMO_OK;
      }
      if (g.N != 64 * NR || geglu || g.colstats) return MIMO_EINVAL;
      g.tiles_n = 1;
      const int64_t nwg = (g.M + 127) / 128;
      if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
      g.ntiles = (unsigned)nwg;
      hipLaunchKernelGGL((gemm_dense_persist_kernel<DT, NR, 2, 4, 4, 1>), dim3((unsigned)(nwg < cus_() ? nwg : cus_())), dim3(512), 0, st, g);
      MIMO_LAUNCH_CHECK();
      return MIMO_OK;
    } else {
      return MIMO_EINVAL;
    }
  }
  // Split-K: a long reduction over few output tiles (the 8x8-level convolutions: M = 3072, K = 11520) leaves most of
  // the chip idle.  Use XL8 tiles, split K over gridDim.y, and let a second tiny launch reduce the fp32 partials
  // (fixed order) and apply the epilogue.  Needs the caller's workspace.
  const int cus = cus_();
  if (forced == 0 && allow_splitk && !geglu && g.ws && NR == 5) {
    const int64_t nt = m256 * tn_xl;
    int64_t sk = nt > 0 ? cus / nt : 0;
    if (sk > g.nkt / 16) sk = g.nkt / 16;
    if (sk > 8) sk = 8;
    const size_t need = (size_t)sk * (size_t)g.M * (size_t)g.N * sizeof(float);
    if (sk >= 2 && nt * 2 <= cus && need <= g.ws_bytes) {
      GemmArgs gp = g;
      gp.tiles_n = tn_xl;
      gp.out = g.ws; gp.ldo = g.N; gp.bias = nullptr; gp.img_bias = nullptr; gp.res = nullptr; gp.out_scale = 1.f;
      gp.flags = (g.flags & 0xffff0000u) | MIMO_EPI_OUT_F32;
      hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, 2, 4, 2, 8>), dim3((unsigned)nt, (unsigned)sk), dim3(512), 0, st, gp);
      MIMO_LAUNCH_CHECK();
      int64_t rb = (g.M * (g.N >> 2) + 255) / 256;
      if (rb > 4096) rb = 4096;
      hipLaunchKernelGGL(splitk_reduce_kernel<DT>, dim3((unsigned)rb), dim3(256), 0, st, g, (int)sk);
      MIMO_LAUNCH_CHECK();
      return MIMO_OK;
    }
  }
  if (cfg == 4) {  // XL8: the XL tile on 8 waves (2 x 4), each wave (16 MT) x 16NR with a 256-register budget
    // Tile height 256 | 192 | 128 rows (MT = 8 | 6 | 4): with one block per CU the launch runs in rounds of `cus`
    // tiles, so pick the height that minimises rounds x (rows + per-tile overhead) — e.g. M = 12288, N = 1280 is
    // 192 tiles of 256 rows (75 % of the chip for one round) but exactly 256 tiles of 192 rows.
    g.tiles_n = tn_xl;
    auto cost = [&](int bm) {
      const int64_t nt = ((g.M + bm - 1) / bm) * tn_xl;
      return ((nt + cus - 1) / cus) * (int64_t)(bm + 24);
    };
    int bm = tn.bm;
    if (bm != 256 && bm != 192 && bm != 128) {
      bm = 256;
      if (cost(192) < cost(bm)) bm = 192;
      if (cost(128) < cost(bm)) bm = 128;
    }
    const int64_t nwg = ((g.M + bm - 1) / bm) * tn_xl;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    // persistent blocks pay off (measured +2..7 %) for short reductions over many rounds of tiles — the level-0 linears;
    // with 2-3 exactly filled rounds or long K the plain launch is faster (tools/microbench.py --ab MIMO_GEMM_PERSIST=0)
    if (MODE == 0 && tn.persist && !g.colstats && !folded && bm == 256 && nwg >= 3 * (int64_t)cus && g.nkt >= 2 && g.nkt <= 20) {
      g.ntiles = (unsigned)nwg;
      hipLaunchKernelGGL((gemm_dense_persist_kernel<DT, NR, 2, 4, 8>), dim3((unsigned)cus), dim3(512), 0, st, g);
      MIMO_LAUNCH_CHECK();
      return MIMO_OK;
    }
    if (bm == 256) MIMO_LAUNCH_GK(2, 4, 2, 8, nwg, 512);
    else if (bm == 192) MIMO_LAUNCH_GK(2, 4, 2, 6, nwg, 512);
    else MIMO_LAUNCH_GK(2, 4, 2, 4, nwg, 512);
  } else if (cfg == 3) {
    g.tiles_n = tn_xl;
    const int64_t nwg = m256 * tn_xl;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    if constexpr (NR == 4) {
      if (MODE == 0 && tn.persist && !folded && nwg >= 3 * (int64_t)cus && g.nkt >= 2) {  // GEGLU: +6 % at every level
        g.ntiles = (unsigned)nwg;
        hipLaunchKernelGGL((gemm_dense_persist_kernel<DT, NR, 4, 4, 4>), dim3((unsigned)cus), dim3(1024), 0, st, g);
      } else {
        MIMO_LAUNCH_GK(4, 4, 2, 4, nwg, 1024);
      }
    }
  } else if (cfg == 2) {
    g.tiles_n = tn_s;
    const int64_t nwg = m256 * tn_s;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, 4, 2, 3, 4>), dim3((unsigned)nwg), dim3(512), 0, st, g);
  } else {
    g.tiles_n = tn_s;
    const int64_t nwg = ((g.M + 127) / 128) * tn_s;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    // Small-M dense launches (the 8 x 8 level: M = 3072, N = 1280 -> 192 blocks of 128 x 160, one per CU on three quarters of
    // the chip, each a serial chain of K / 64 round trips): 64-row tiles double the blocks (two per CU: 57 KB of LDS each), i.e.
    // the bytes in flight per CU, which is what a latency-bound launch is short of.  A function of (M, N) only; the K order of
    // every output element is unchanged, so the result is bit-identical to the 128-row tile's.
    if constexpr (MODE == 0) {
      if (tune_env("MIMO_GEMM_BM64", GEMM_BM64_DEFAULT) && nwg <= cus && !geglu) {
        const int64_t nwg64 = ((g.M + 63) / 64) * tn_s;
        MIMO_LAUNCH_GK(2, 2, 2, 2, nwg64, 256);
        MIMO_LAUNCH_CHECK();
        return MIMO_OK;
      }
    }
    MIMO_LAUNCH_GK(2, 2, 2, 4, nwg, 256);
  }
  (void)geglu;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

// MIMO_GEMM_8P (tune build; the shipped default is the constant below): 1 = dense GEMMs with an even number of K-tiles
// and enough 256 x 256 tiles take the 8-phase kernel
constexpr int GEMM_8P_DEFAULT = 1;

// Column-slot width of mimo_epilogue_ext.row_stats = the 16 NR columns one wave owns in the kernel family a producer of
// width N runs on — chosen from N alone (never from M), so a row's partial sums do not depend on the batch:
// 64 (gemm8_kernel, or the NR = 4 tiles when the row count is too small for it) where 256-wide tiles fit N, else 80 (NR = 5).
int row_slot_width(int N) {
  const int tn8 = (N + 255) / 256;
  if (tn8 * 256 - N <= N / 12 && N % 64 == 0) return 64;
  if (N % 160 == 0) return 80;
  return N % 64 == 0 ? 64 : 0;
}

// Slots per row of a folded LayerNorm of width N, or 0 where the fold does not apply: the producer's row-side epilogue has no
// column mask, so N must be a whole number of the WIDEST tile its kernel family uses (4 slots: 256 columns for the 64-wide
// slots, 320 for the 80-wide ones — 480 / 800 / 960 / 1920 would let a wave past column N write into the next row); the
// consumer holds at most 2 * ROW_STAT_LOADS slots (N = 2560 has 40).
int row_stat_slots_of(int N) {
  const int w = N > 0 ? row_slot_width(N) : 0;
  if (!w || N % (4 * w) != 0) return 0;
  const int slots = N / w;
  return (slots & 1) == 0 && slots <= 2 * ROW_STAT_LOADS ? slots : 0;
}

template <int DT, int MODE>
int launch(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
  if constexpr (MODE == 0) {
    const int p8 = tune_env("MIMO_GEMM_8P", GEMM_8P_DEFAULT);
    const int64_t tn8 = (g.N + 255) / 256, nt8 = ((g.M + 255) / 256) * tn8;
    // (N = 640 is 2.5 tiles of 256: a sixth of the columns would be padding — those shapes keep the 320-wide tiles)
    const bool n_fits = tn8 * 256 - g.N <= g.N / 12;
    if (p8 && !g.colstats && !g.ln_out && (g.K % 128) == 0 && n_fits && nt8 >= 128 && nt8 <= 0x7fffffff && g.M < 0x7fffffff) {
      g.tiles_n = (int)tn8;
      g.ntiles = (unsigned)tune_env("MIMO_G8_GM", G8_GM);
      if (g.row_half || g.a_row_stats) hipLaunchKernelGGL((gemm8_kernel<DT, 2>), dim3((unsigned)nt8), dim3(512), 0, st, g);
      else hipLaunchKernelGGL((gemm8_kernel<DT>), dim3((unsigned)nt8), dim3(512), 0, st, g);
      MIMO_LAUNCH_CHECK();
      return MIMO_OK;
    }
  }
  if (g.row_half) return row_slot_width(g.N) == 80 ? launch_nr<DT, MODE, 5>(g, st) : launch_nr<DT, MODE, 4>(g, st);
  // NR = 5 (BN = 160) divides every SD1.5 width (320/640/960/1280/1920/2560); NR = 4 otherwise
  if (!geglu && (g.N % 160 == 0)) return launch_nr<DT, MODE, 5>(g, st);
  return launch_nr<DT, MODE, 4>(g, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

static int check_ext(const mimo_epilogue_ext* e, GemmArgs& g, int64_t M, int N, int K, unsigned flags) {
  g.colstats = nullptr; g.ln_gamma = g.ln_beta = g.ln_pe = nullptr; g.ln_out = nullptr;
  g.ln_eps = 0.f; g.ln_rows_per_frame = 1; g.ln_pe_frames = 1;
  g.row_half = nullptr; g.row_stats = nullptr; g.a_row_stats = nullptr; g.a_colsum = nullptr; g.a_slots = 0; g.a_eps = 0.f;
  if (!e) return MIMO_OK;
  if (e->row_half || e->row_stats) {   // producer half of a folded LayerNorm: fp32 result + half copy + row statistics
    if (!e->row_half || !e->row_stats || e->colstats || e->ln_out || !(flags & MIMO_EPI_OUT_F32) ||
        (flags & (MIMO_EPI_SILU | MIMO_EPI_GEGLU)) || row_stat_slots_of(N) == 0 || M >= 0x7fffffffLL / ((int64_t)N * 2) ||
        !aligned16(e->row_half) || !aligned16(e->row_stats))
      return MIMO_EINVAL;
    g.row_half = e->row_half; g.row_stats = e->row_stats;
  }
  if (e->a_row_stats || e->a_colsum) {  // consumer half
    // (a_slots must be the producer's slot count for THIS K: statistics of another width would be read misaligned)
    if (!e->a_row_stats || !e->a_colsum || e->a_slots <= 0 || e->a_slots != row_stat_slots_of(K) || e->colstats || e->ln_out ||
        M >= 0x7fffffffLL / ((int64_t)e->a_slots * 8) || !aligned16(e->a_row_stats) || !aligned16(e->a_colsum))
      return MIMO_EINVAL;
    g.a_row_stats = e->a_row_stats; g.a_colsum = e->a_colsum; g.a_slots = e->a_slots; g.a_eps = e->a_eps;
  }
  if (e->colstats) {
    if ((M & 31) || (N & 3) || (flags & MIMO_EPI_GEGLU) || !aligned16(e->colstats)) return MIMO_EINVAL;
    g.colstats = e->colstats;
  }
  if (e->ln_out) {
    if (!e->ln_gamma || !e->ln_beta || e->colstats || !aligned16(e->ln_out)) return MIMO_EINVAL;
    if (e->ln_pe && (e->ln_rows_per_frame <= 0 || e->ln_pe_frames <= 0 || (e->ln_rows_per_frame % 128))) return MIMO_EINVAL;
    g.ln_gamma = e->ln_gamma; g.ln_beta = e->ln_beta; g.ln_pe = e->ln_pe; g.ln_out = e->ln_out;
    g.ln_eps = e->ln_eps;
    g.ln_rows_per_frame = e->ln_pe ? e->ln_rows_per_frame : 1;
    g.ln_pe_frames = e->ln_pe ? e->ln_pe_frames : 1;
  }
  return MIMO_OK;
}

extern "C" int mimo_gemm_ext(int dtype, const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                             int64_t M, int N, int K, const float* bias, const float* img_bias,
                             int64_t img_bias_ld, int64_t rows_per_img, const void* residual, int64_t ldr,
                             float out_scale, unsigned flags, void* workspace, size_t workspace_bytes,
                             const mimo_epilogue_ext* ext, void* stream) {
  if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0) return MIMO_EINVAL;
  if ((K & 7) || (lda & 7) || (N & 3) || (ldo & 3) || !aligned16(A) || !aligned16(W) || !aligned16(out))
    return MIMO_EINVAL;
  if ((flags & MIMO_EPI_GEGLU) && (N & 31)) return MIMO_EINVAL;
  if (flags & ~(MIMO_EPI_SILU | MIMO_EPI_GEGLU | MIMO_EPI_OUT_F32 | MIMO_EPI_RES_F32 | MIMO_EPI_NO_SPLITK)) return MIMO_EINVAL;
  if (residual && (ldr & 3)) return MIMO_EINVAL;
  if (img_bias && (rows_per_img <= 0 || (img_bias_ld & 3))) return MIMO_EINVAL;
  GemmArgs g{};
  g.A = (const uint16_t*)A; g.A2 = nullptr; g.W = (const uint16_t*)W; g.out = out;
  g.bias = bias; g.img_bias = img_bias; g.res = residual;
  g.lda = lda; g.ldw = K; g.ldo = ldo; g.ldr = ldr; g.M = M; g.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
  g.ldib = img_bias_ld > 0 ? img_bias_ld : N;
  g.N = N; g.K = K; g.out_scale = out_scale; g.flags = flags;
  g.ws = aligned16(workspace) ? workspace : nullptr; g.ws_bytes = workspace_bytes;
  g.nkt = (K + BK - 1) / BK;
  if (int rc = check_ext(ext, g, M, N, K, flags)) return rc;
  if (g.ln_out && ((flags & (MIMO_EPI_SILU | MIMO_EPI_GEGLU)) || ldo != N || (residual && ldr != N))) return MIMO_EINVAL;
  const int64_t ab = ((M - 1) * lda + K) * 2, wb = (int64_t)N * K * 2;
  if (ab >= 0x80000000LL || wb >= 0x80000000LL) return MIMO_EINVAL;  // 32-bit offsets; 2 GiB keeps OOBA + soffset out of range
  g.a_bytes = (unsigned)ab; g.a2_bytes = 0; g.w_bytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  // level-0 token linears (K = 320, plain half output or GEGLU): A in registers, W streamed (gemm_stream.hip)
#ifndef MIMO_NO_STREAM
#define MIMO_NO_STREAM 0
#endif
  if (!MIMO_NO_STREAM && tune_env("MIMO_GEMM_STREAM", 1) && mimo_stream::supported(M, N, K, (flags & MIMO_EPI_NO_SPLITK) != 0) && !residual && !img_bias && !g.colstats && !g.ln_out && !g.row_half && !g.a_row_stats &&
      !(flags & (MIMO_EPI_SILU | MIMO_EPI_OUT_F32)) && out_scale == 1.f && (lda & 7) == 0 && (ldo & 3) == 0 &&
      aligned16(A) && aligned16(W) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0) {
    mimo_stream::Args a{};
    a.A = g.A; a.W = g.W; a.out = out; a.bias = bias; a.lda = lda; a.ldo = ldo; a.M = M; a.N = N;
    a.geglu = (flags & MIMO_EPI_GEGLU) ? 1 : 0;
#ifdef MIMO_TUNE
    a.dbg = tune_env("MIMO_GEMM_TRACE", 0) ? trace_buf() : nullptr;
    a.ablate = tune_env("MIMO_STREAM_ABLATE", 0);
#endif
    return mimo_stream::launch(dtype, a, cus_(), st);
  }
  if (dtype == MIMO_F16) return launch<MIMO_F16, 0>(g, st);
  if (dtype == MIMO_BF16) return launch<MIMO_BF16, 0>(g, st);
  return MIMO_EDTYPE;
}

extern "C" int mimo_gemm(int dtype, const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                         int64_t M, int N, int K, const float* bias, const float* img_bias,
                         int64_t img_bias_ld, int64_t rows_per_img, const void* residual, int64_t ldr,
                         float out_scale, unsigned flags, void* workspace, size_t workspace_bytes, void* stream) {
  return mimo_gemm_ext(dtype, A, lda, W, out, ldo, M, N, K, bias, img_bias, img_bias_ld, rows_per_img, residual, ldr,
                       out_scale, flags, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mimo_conv2d_ext(int dtype, const void* in, const void* in2, const void* W, void* out,
                               const mimo_conv_params* p, const float* bias, const float* img_bias,
                               const void* residual, float out_scale, unsigned flags, void* workspace,
                               size_t workspace_bytes, const mimo_epilogue_ext* ext, void* stream) {
  if (!in || !W || !out || !p) return MIMO_EINVAL;
  if (p->n <= 0 || p->Cin <= 0 || (p->Cin & 7) || (p->Cout & 3) || p->Cout <= 0) return MIMO_EINVAL;
  if (!(p->ksize == 1 || p->ksize == 3) || !(p->stride == 1 || p->stride == 2)) return MIMO_EINVAL;
  if (p->Cin2 < 0 || (p->Cin2 & 7) || (p->Cin2 > 0 && !in2)) return MIMO_EINVAL;
  if ((p->Hup > 0) != (p->Wup > 0)) return MIMO_EINVAL;
  if (flags & ~(MIMO_EPI_SILU | MIMO_EPI_OUT_F32 | MIMO_EPI_RES_F32 | MIMO_EPI_NO_SPLITK)) return MIMO_EINVAL;
  if (!aligned16(in) || !aligned16(W) || !aligned16(out) || (in2 && !aligned16(in2))) return MIMO_EINVAL;
  if (ext && (ext->ln_out || ext->row_half || ext->row_stats || ext->a_row_stats || ext->a_colsum))
    return MIMO_EINVAL;  // the LayerNorm side outputs / folded LayerNorm exist for dense GEMMs only
  // thin-input layers (pose guider, VAE conv_in): direct convolution, see thinconv.hip
  if (tune_env("MIMO_THIN_CONV", 1) && !in2 && p->Cin2 == 0 && !img_bias && !residual && !(ext && ext->colstats) && p->Hup == 0 &&
      !(flags & ~(MIMO_EPI_SILU | MIMO_EPI_OUT_F32 | MIMO_EPI_NO_SPLITK))) {
    const int64_t ib = (int64_t)p->n * p->Hin * p->Win * p->Cin * 2;
    const int64_t ob = (int64_t)p->n * p->Hout * p->Wout * p->Cout * ((flags & MIMO_EPI_OUT_F32) ? 4 : 2);
    if (mimo_thin::supported(p->Cin, p->Cout, p->ksize, p->stride, ib, ob) && (p->ksize == 3 || (p->pad_t == 0 && p->pad_l == 0))) {
      mimo_thin::Args a{};
      a.in = (const uint16_t*)in; a.W = (const uint16_t*)W; a.out = out; a.bias = bias;
      a.n = p->n; a.Hin = p->Hin; a.Win = p->Win; a.Cin = p->Cin; a.Hout = p->Hout; a.Wout = p->Wout; a.Cout = p->Cout;
      a.ksize = p->ksize; a.stride = p->stride; a.pad_t = p->pad_t; a.pad_l = p->pad_l; a.ldw = (int64_t)p->ksize * p->ksize * p->Cin;
      a.out_scale = out_scale; a.flags = flags & (MIMO_EPI_SILU | MIMO_EPI_OUT_F32);
      a.in_bytes = (unsigned)ib; a.w_bytes = (unsigned)((int64_t)p->Cout * a.ldw * 2); a.out_bytes = (unsigned)ob;
      return mimo_thin::launch(dtype, a, cus_(), (hipStream_t)stream);
    }
  }
  GemmArgs g{};
  g.A = (const uint16_t*)in; g.A2 = (const uint16_t*)in2; g.W = (const uint16_t*)W; g.out = out;
  g.bias = bias; g.img_bias = img_bias; g.res = residual;
  g.N = p->Cout; g.K = p->ksize * p->ksize * p->Cin + p->Cin2;
  g.ldw = g.K; g.ldo = p->Cout; g.ldr = p->Cout; g.lda = 0;
  g.M = (int64_t)p->n * p->Hout * p->Wout;
  g.rows_per_img = (int64_t)p->Hout * p->Wout * (p->imgs_per_bias_row > 0 ? p->imgs_per_bias_row : 1);
  g.ldib = p->img_bias_ld > 0 ? p->img_bias_ld : p->Cout;
  if (g.ldib & 3) return MIMO_EINVAL;
  g.out_scale = out_scale; g.flags = flags;
  g.ws = aligned16(workspace) ? workspace : nullptr; g.ws_bytes = workspace_bytes;
  g.Hin = p->Hin; g.Win = p->Win; g.Cin = p->Cin; g.Hout = p->Hout; g.Wout = p->Wout;
  g.ks = p->ksize; g.stride = p->stride; g.pad_t = p->pad_t; g.pad_l = p->pad_l;
  g.Hup = p->Hup; g.Wup = p->Wup; g.Cin2 = p->Cin2;
  g.sh = p->Hup > 0 ? (float)p->Hin / (float)p->Hup : 1.f;
  g.sw = p->Wup > 0 ? (float)p->Win / (float)p->Wup : 1.f;
  g.chunks1 = (p->Cin + BK - 1) / BK;
  g.chunks2 = (p->Cin2 + BK - 1) / BK;
  g.nkt = p->ksize * p->ksize * g.chunks1 + g.chunks2;
  if (int rc = check_ext(ext, g, g.M, g.N, g.K, flags)) return rc;
  {
    const int64_t ab = (int64_t)p->n * p->Hin * p->Win * p->Cin * 2, a2b = g.M * p->Cin2 * 2, wb = (int64_t)g.N * g.K * 2;
    if (ab >= 0x80000000LL || a2b >= 0x80000000LL || wb >= 0x80000000LL) return MIMO_EINVAL;  // see OOBA
    g.a_bytes = (unsigned)ab; g.a2_bytes = (unsigned)a2b; g.w_bytes = (unsigned)wb;
  }
  const bool ups = p->Hup > 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16) return ups ? launch<MIMO_F16, 2>(g, st) : launch<MIMO_F16, 1>(g, st);
  if (dtype == MIMO_BF16) return ups ? launch<MIMO_BF16, 2>(g, st) : launch<MIMO_BF16, 1>(g, st);
  return MIMO_EDTYPE;
}

extern "C" int mimo_conv2d(int dtype, const void* in, const void* in2, const void* W, void* out,
                           const mimo_conv_params* p, const float* bias, const float* img_bias,
                           const void* residual, float out_scale, unsigned flags, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return mimo_conv2d_ext(dtype, in, in2, W, out, p, bias, img_bias, residual, out_scale, flags, workspace,
                         workspace_bytes, nullptr, stream);
}

extern "C" int mimo_row_stat_slots(int N) { return row_stat_slots_of(N); }

extern "C" int mimo_version(void) { return 5; }

extern "C" size_t mimo_workspace_bytes(void) { return (size_t)8 * ((size_t)1 << 22) * sizeof(float); }

// Kept for tools/microbench.py: the tune build re-reads its knobs at every launch, the shipped build has none.
extern "C" int mimo_reload_tuning(void) { return MIMO_OK; }
