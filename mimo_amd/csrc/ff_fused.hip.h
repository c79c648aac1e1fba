// ff_fused.hip.h — argument block, LDS geometry and small helpers shared by the fused transformer-block kernels
// (ff_fused.hip: 8 waves x 256 registers; ff_tail4.hip: 4 waves x 512 registers).
#pragma once
#include "common.hip.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct FFArgs {
  const uint16_t* A;    // half [M, lda], C columns
  const uint16_t* W1;   // half [8C, C] GEGLU-packed (16 value rows | 16 gate rows blocks)
  const uint16_t* W2;   // half [C, 4C], K axis permuted inside every 32-block (pack_ff2_kperm)
  const float* b1;      // [8C] packed like W1
  const float* b2;      // [C]
  const float* res;     // fp32 [M, ldr]
  uint16_t* out;        // half [M, ldo]                                         (MODE 0)
  int64_t lda, ldr, ldo, M;
  // MODE >= 1: the block's output projection folded in:  out32 = x + (res + FF(A)) @ Wp^T + bp
  const uint16_t* Wp;   // half [C, C]: rows in tile order (tile q = columns 32q..32q+31 | 160+32q..160+32q+31), K axis permuted
  const float* bp;      // [C]
  const float* x;       // fp32 [M, ldx]: the block input (the residual of proj_out)
  float* out32;         // fp32 [M, ldo32]
  int64_t ldx, ldo32;
  float* colstats;      // MODE >= 1, optional: fp32 [M / 32, 2, C] GroupNorm column statistics of out32 per 32-row slab (as mimo_gemm_ext)
  // MODE = 2: the attention output projection and the LayerNorm in front of the feed-forward folded in as well:
  //   y = res + A @ Wo^T + bo (+ img_bias[row / rows_per_img]);  n = LayerNorm(y) * gamma + beta;  out32 = x + (y + FF(n)) @ Wp^T + bp
  // (A = the attention output, res = the stream before the attention; W1 then carries the K permutation of pack_ff2_kperm)
  // W1 is then ONE stream of 50 tiles of 64 rows: [Wo (rows in tile order like Wp, K axis natural) | W1 | Wp]
  const float* bo;      // [C]
  const float* img_bias;  // fp32 [nimg, ldib] or null
  const float* ln_gamma;  // [C]
  const float* ln_beta;   // [C]
  int64_t ldib, rows_per_img;
  float ln_eps;
  // MODE 3 / 4 (block head): W1 = ONE stream of 20 tiles of 64 rows [Wi (rows in tile order, K natural) | Wqkv (rows natural,
  // K permuted)]; A (half, MODE 3) or x + gn_ab (MODE 4) = the operand of the first projection; res optional; out32 = y;
  // out = qkv (half [M, ldo], 3C columns)
  const float* gn_ab;     // MODE 4: fp32 [nimg, 2, C] GroupNorm folded to x * a + b per (image, channel); rows_per_img >= 128
  const float* ln_pe;     // optional fp32 [ln_pe_frames, C] added to the LayerNorm output, row = (m / ln_rows_per_frame) % frames
  int64_t ln_rows_per_frame;
  int ln_pe_frames;
#ifdef MIMO_TUNE
  unsigned long long* dbg;  // phase trace (tools/ff_trace.py) or null
  int ablate;               // 1: no DMAs (stale tiles); 2: no MFMA phases (stream + barriers only); 3: GELU -> identity
#endif
};

#ifdef MIMO_TUNE
// two tracers of block 0: thread 0 (wave 0: row group 0, column half 0) fills entries [0, 2000), thread 256 (wave 4: column
// half 1, the DMA issuer) [2000, 4000)
#define FF_TRACE(g, idx, tag)                                                                                      \
  do {                                                                                                             \
    if ((g).dbg && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (idx) < 2000u)                                   \
      (g).dbg[(threadIdx.x >> 8) * 2000u + (idx)++] =                                                              \
          ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull);              \
  } while (0)
#define FF_ABLATE(g, n) ((g).ablate == (n))
#else
#define FF_TRACE(g, idx, tag) do { (void)(idx); } while (0)
#define FF_ABLATE(g, n) false
#endif

template <int V>
struct ICf {
  static constexpr int value = V;
};

constexpr int C = 320, KS = C / 32, HID = 4 * C, NSTEP = HID / 32;   // 40 steps per panel
constexpr int ROWB1 = C * 2;                  // bytes of a W1 row
constexpr int W1_TILE = 64 * ROWB1;           // 40 KB
constexpr int W2_TILE = C * 64;               // 320 rows x 32 k x 2 B = 20 KB
constexpr int STAGE = W1_TILE + W2_TILE;      // 60 KB
constexpr int BIAS_OFF = 2 * STAGE;           // b1 (8C floats), b2, bp, bo, LN gamma, LN beta (C floats each)
constexpr int XCH_OFF = BIAS_OFF + (8 * C + 5 * C) * 4;
constexpr int NTAIL = C / 64;                 // W tiles of the folded output projection
constexpr int LDS_BYTES = XCH_OFF + 2 * 8 * 1024;   // hidden-chunk exchange, double-buffered by step parity
constexpr int BM = 128;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");

// An offset that must stay ONE register inside the panel loop: everything derived from it by adding constants then folds into
// the instructions' immediate offsets.  (Left visible to the optimiser, every `offset + constant` is loop-invariant, gets
// hoisted out of the panel loop as a value of its own and is parked in scratch across the main loop.)
__device__ __forceinline__ unsigned pinned(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

}  // namespace
