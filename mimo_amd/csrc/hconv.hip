// hconv.hip — halo-tiled 3x3 convolution whose INPUT normalisation is part of the operand path (gfx950).
//
//   out = epilogue( conv3x3( silu?( x * a + b ) ) )        a, b per (image, channel) = GroupNorm folded to an affine
//
// replaces the pair  gn_apply_kernel (read fp32, write half)  +  gemm_kernel<MODE 1|2> (implicit GEMM that re-fetches
// every input pixel nine times through L2)  of  ResnetBlock3D / ResnetBlock2D  (src/models/resnet.py:217-247, the VAE
// res blocks) and the cast + nearest-x2 gather + conv of Upsample3D (src/models/resnet.py:31-76).
//
// Structure.  A block owns a 16 x 16 pixel tile of ONE image x BN output channels (8 waves, each MT tile rows x 16 NR
// columns as 16x16x32 MFMAs, fp32 accumulators).  Per 32-channel sub-chunk the (16+2)^2 halo patch of the fp32 input is
// fetched ONCE (LDS-DMA into a staging area, 16 KB per pass), normalised + SiLU'd + rounded to half by the VALU once
// (3 passes of 8 values per thread: ~60 VALU instructions per pass beside 80 MFMAs per wave and step) and written to an
// LDS patch image [18 rows][24 px pitch][32 ch] — the nine taps are nine fragment-base offsets into that image, so
// only the weights stream through the DMA ring (2 slots of two tap-steps each = 2 x BN x 128 B).  The two patch images
// (sub-chunks a / b of a 64-channel chunk) are written while the other one is being multiplied: no stall at chunk
// boundaries, no half-precision intermediate in HBM, one read of the fp32 tensor (x 1.27 for the halo).
//
// K order: 64-channel chunk outer; inside it 18 tap-steps (sub-chunk a: taps 0..8, sub-chunk b: taps 0..8), two per
// barrier step (K = 2 x 32 = the 64-deep step of gemm_conv.hip).  The order is a function of the layer only: a frame's
// result does not depend on the batch it is launched with.
//
// LDS images are XOR-swizzled on the 16-byte slot (slot ^= 2 * bit 2 of the row/pixel index): with 64-byte rows the
// ds_read_b128 fragment reads of 16 consecutive rows are bank-conflict-free; the W tiles get the swizzle on the DMA's
// SOURCE address (the DMA writes lane-linear), the patch image on the ds_write address.
// Every vector-memory instruction of the main loop is inline asm and counted by hand (s_waitcnt vmcnt): hipcc drains
// the DMA queue before LDS reads it cannot prove independent.
//
// Side outputs: the half cast of the raw input (the operand of conv2's fused 1x1 shortcut), and GroupNorm statistics of the
// tensor the launch writes (see HArgs::tstats), merged by mimo_group_norm_stats_slabs: the norm that consumes the output
// makes no statistics pass over HBM.
#include "common.hip.h"

// Compile-time experiment knobs (tools/hconv_variants.py builds one small library per setting and times them interleaved
// in one process; the shipped library is built with the defaults below).
#ifndef HCONV_LATE_CONSUME   // 1: waves 4-7 transform their patch pass AFTER the second half of a step (their SIMD partners
#define HCONV_LATE_CONSUME 0 //    0-3 do it between the halves): the two VALU phases of a SIMD do not coincide
#endif
#ifndef HCONV_FENCE          // 1: sched_barrier fences around the patch work (keeps fragment live ranges inside a half)
#define HCONV_FENCE 1
#endif
#ifndef HCONV_SETPRIO        // 1: s_setprio 1 around every MFMA cluster
#define HCONV_SETPRIO 0
#endif
#ifndef HCONV_PRO            // 1: prologue with the first weight tile in flight beside the first patch passes (one barrier less)
#define HCONV_PRO 1
#endif
#ifndef HCONV_ABLATE         // timing experiments (results are wrong): 1 no patch VALU, 2 no MFMA, 4 no weight DMA, 8 no patch DMA
#define HCONV_ABLATE 0
#endif

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V>
struct IC {
  static constexpr int value = V;
};

struct HArgs {
  const float* x1;        // fp32 [n, Hs, Ws, C1]
  const float* x2;        // fp32 [n, Hs, Ws, C2] (virtual channel concat) or null
  const float* ab;        // fp32 [n][2][C]: y = x * a + b; null = plain cast
  const uint16_t* W;      // half16 [N][ldw], K index = tap * C + c
  float* out;             // fp32 [n, H, W, N]
  const float* bias;      // [N] or null
  const float* img_bias;  // [n / imgs_per_bias_row][ldib] or null
  const float* res;       // fp32 [n, H, W, N] or null
  uint16_t* raw;          // optional side output: half16 cast of the un-normalised input [n, H, W, C]
  float* tstats;          // optional side output, GroupNorm statistics of `out` as (mean, centred sum of squares) per channel:
                          //   without a residual per 16 x 16 tile [tiles][2][N] (from the accumulators, TSTATS = 1),
                          //   with one per 32-pixel slab = two tile rows [n H W / 32][2][N] (in the store loop, TSTATS = 2)
  int64_t ldw, ldib;
  int C1, C2, n, H, Wd, Hs, Ws, ups, N, tiles_n, tiles_x, tiles_y, silu, epi_silu, imgs_per_bias_row;
  float out_scale;
  unsigned w_bytes;
};

constexpr unsigned OOBA = 0x80000000u;  // every descriptor is < 2 GiB: OOBA (+ soffset) is out of range -> zero fill / dropped store
constexpr int PB = 18 * 24 * 64;        // bytes of one patch image: 18 rows x 24-pixel pitch x 32 half channels
constexpr int STG = 16384;              // staging bytes of one patch pass: 512 threads x 2 x 16 B

template <int DT, int NR, int WM, int WN, int ABMAX, bool NORM, bool SIDE, int TSTATS>
__global__ __launch_bounds__(512, 2) void hconv_kernel(const HArgs g) {
  static_assert(WM * WN == 8, "8 waves");
  constexpr int MT = 16 / WM;       // tile rows (= 16-pixel MFMA column tiles) per wave
  constexpr int BN = 16 * NR * WN;  // output channels per block
  constexpr int NWD = BN / 64;      // weight DMAs per wave and barrier step (2 tap-steps x BN rows x 64 B / 1 KB / 8 waves)
  constexpr int WTS = BN * 64;      // bytes of one tap-step of weights
  constexpr int WSLOT = 2 * WTS;
  constexpr int OFF_PA = 0, OFF_PB = PB, OFF_AB = 2 * PB, OFF_W = OFF_AB + ABMAX, OFF_STG = OFF_W + 2 * WSLOT;
  constexpr int LDS_BYTES = OFF_STG + STG;
  static_assert(2 * WSLOT + STG >= 3 * STG, "the prologue stages three passes in the weight ring");
  static_assert(LDS_BYTES <= 163840, "LDS");
  // ONE LDS object (a second __shared__ array would make hipcc drain vmcnt before fragment reads)
  __shared__ __attribute__((aligned(16))) uint4 smem[LDS_BYTES / 16];
  char* const lds = reinterpret_cast<char*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = (int)(wave_u / (unsigned)WN), wn = (int)(wave_u % (unsigned)WN);
  const int lg = lane >> 4, li = lane & 15;
  const unsigned ts = wave_u >> 2;  // which tap-step of a barrier step this wave's weight DMAs fetch; also: issues them late
  const int q = tid & 3;            // 8-channel group of the 32-channel sub-chunk this thread transforms

  // ---- block -> (image, tile, channel tile) ----
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = (int)(L % (unsigned)g.tiles_n);
  unsigned t_ = L / (unsigned)g.tiles_n;
  const int tx = (int)(t_ % (unsigned)g.tiles_x);
  t_ /= (unsigned)g.tiles_x;
  const int ty = (int)(t_ % (unsigned)g.tiles_y);
  const int img = (int)(t_ / (unsigned)g.tiles_y);
  const int Y0 = ty * 16, X0 = tx * 16, N0 = tile_n * BN;
  const int C = g.C1 + g.C2;
  const int nch = (C + 63) >> 6;

  // ---- patch items: pass p, thread t -> item e = 512 p + t = (patch pixel e / 4, channel group e % 4) ----
  // One register per pass: source pixel index (21 bits; pixels outside the image get index Hs*Ws = the first offset
  // beyond the image's buffer descriptor, so the DMA zero-fills them without a select) | patch slot (16-byte units) << 21.
  const unsigned src_px_u = (unsigned)(g.Hs * g.Ws);
  unsigned pk[3];
  unsigned imask = 0;  // bit p: interior pixel (raw side output)
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const int e = p * 512 + tid;
    const int pp = e >> 2;
    const bool ex = pp < 324;
    const int prow = pp / 18, pcol = pp - prow * 18;
    const int y = Y0 + prow - 1, x = X0 + pcol - 1;
    const bool ok = ex && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.Wd;
    const int ys = g.ups ? (y >> 1) : y, xs = g.ups ? (x >> 1) : x;
    const int lp = prow * 24 + pcol;
    const unsigned slot = (unsigned)(lp * 4 + (q ^ (((lp >> 2) & 1) << 1)));
    pk[p] = (ok ? (unsigned)(ys * g.Ws + xs) : src_px_u) | (slot << 21);
    if (ok && prow >= 1 && prow <= 16 && pcol >= 1 && pcol <= 16) imask |= 1u << p;
  }

  // Loop-invariant values DERIVED from these few registers (DMA / store offsets, LDS addresses) are re-derived where they
  // are used: step() passes the registers through an empty asm, which makes them opaque to the optimiser.  Left alone,
  // LICM hoists a dozen derived addresses out of the 9-step loop and the register allocator parks them in scratch —
  // and every scratch reload is a vector-memory load hipcc waits for with vmcnt(0), i.e. it drains the DMA queue.
  unsigned tid16 = (unsigned)tid * 16u;

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const int64_t src_px = (int64_t)g.Hs * g.Ws;
  const i32x4 rX1 = make_rsrc(g.x1 + (int64_t)img * src_px * g.C1, (unsigned)(src_px * g.C1 * 4));
  const i32x4 rX2 = make_rsrc(g.x2 ? g.x2 + (int64_t)img * src_px * g.C2 : nullptr, g.x2 ? (unsigned)(src_px * g.C2 * 4) : 0u);
  const i32x4 rW = make_rsrc(g.W, g.w_bytes);
  const int64_t out_px = (int64_t)g.H * g.Wd;
  const bool side = SIDE && tile_n == 0;
  const i32x4 rRaw = make_rsrc(side ? g.raw + (int64_t)img * out_px * C : nullptr, side ? (unsigned)(out_px * C * 2) : 0u);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];

  // One DMA = 64 lanes x 16 B; lane l lands at lds_base + 16 l.  Inline asm: hipcc must not track it.
  auto sgpr4 = [](const i32x4& r) -> i32x4 {
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane(r.x); rs.y = __builtin_amdgcn_readfirstlane(r.y);
    rs.z = __builtin_amdgcn_readfirstlane(r.z); rs.w = __builtin_amdgcn_readfirstlane(r.w);
    return rs;
  };
  auto dma = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned lds_base) {
    // (readfirstlane: a no-op for values the compiler already keeps in scalar registers; it moves descriptors it
    // decided to park in vector registers back where the instruction needs them)
    const i32x4& rs = r;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rs), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_base))
                 : "memory", "m0");
  };

  // ---- weight DMAs: wave w fetches tap-step ts = w / 4 of a barrier step, rows ((w % 4) NWD + j) * 16 .. + 16 ----
  // lane offset of DMA 0 ; DMA j adds 16 rows through the scalar offset; every
  // row of the tile exists (N % BN == 0, checked by the launcher: the scalar offset is not range checked)
  const unsigned wv0 = ((unsigned)N0 + (wave_u & 3u) * (unsigned)(NWD * 16) + (unsigned)(lane >> 2)) * (unsigned)(g.ldw * 2) +
                       (unsigned)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) << 4);
  const unsigned wrow16 = (unsigned)(g.ldw * 32);  // bytes of 16 weight rows
  // the W tile of (chunk cn, step jn) into ring slot `slot`: this wave's tap-step is idx = 2 jn + ts
  auto issue_w = [&](int jn, int cn, unsigned slot) {
    const int idx = 2 * jn + (int)ts;
    const int sub = idx >= 9 ? 1 : 0, tap = idx - 9 * sub;
    const int cc = cn * 64 + sub * 32;
    i32x4 r = rW;
    r.z = cc < C ? r.z : 0;  // past the last sub-chunk: zero fill (keeps the DMA counts uniform)
    const unsigned soff = cc < C ? (unsigned)(tap * C + cc) * 2u : 0u;
    const unsigned base = smem_base + (unsigned)OFF_W + slot * (unsigned)WSLOT + ts * (unsigned)WTS + (wave_u & 3u) * (unsigned)(NWD * 1024);
    if (HCONV_ABLATE & 4) return;
#pragma unroll
    for (int j = 0; j < NWD; ++j) dma(r, wv0, soff + (unsigned)j * wrow16, base + (unsigned)(j * 1024));
  };

  // ---- patch pass p of the 32-channel sub-chunk at concat channel cc: fp32 -> staging (2 DMAs) ----
  auto issue_patch = [&](int p, int cc, unsigned stg) {
    const bool first = cc < g.C1;
    const int Cx = first ? g.C1 : g.C2;
    const int ccl = first ? cc : cc - g.C1;
    i32x4 r = first ? rX1 : rX2;
    r.z = cc < C ? r.z : 0;
    const unsigned pkp = pk[p];
    const unsigned voff = (pkp & 0x1fffffu) * (unsigned)(Cx * 4) + ((tid16 & 48u) << 1);  // + 32 q
    const unsigned soff = cc < C ? (unsigned)(ccl * 4) : 0u;
    const unsigned base = smem_base + stg + wave_u * 1024u;
    if (HCONV_ABLATE & 8) return;
    dma(r, voff, soff, base);
    dma(r, voff, soff + 16u, base + 8192u);  // (an instruction offset would also shift the LDS address)
  };
  const float* abt = reinterpret_cast<const float*>(lds + OFF_AB);
  constexpr bool norm = NORM, do_silu = NORM;  // (affine without SiLU is not instantiated: no 3x3 convolution on this path needs it)
  // staging -> (affine, SiLU, half) -> patch image `pbuf`; optional raw side output
  auto consume = [&](int p, int cc, unsigned stg, unsigned pbuf) {
    // zero padding is padding of the NORMALISED tensor: pixels outside the image (and channels past the end) become 0
    const unsigned pkp = pk[p];
    const bool ok = (pkp & 0x1fffffu) != src_px_u && cc < C;
    uint32_t w[4], rw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {  // channels 4 hh .. 4 hh + 3 of the thread's 8
      const f32x4 xv = *reinterpret_cast<const f32x4*>(lds + stg + hh * 8192 + tid16);
      f32x4 yv = xv;
      if (norm && !(HCONV_ABLATE & 1)) {
        const unsigned qo = (tid16 & 48u) >> 1;  // 8 q floats
        const f32x4 av = *reinterpret_cast<const f32x4*>(abt + cc + qo + hh * 4);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(abt + C + cc + qo + hh * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) yv[k] = fmaf(xv[k], av[k], bv[k]);
      }
      if (do_silu && !(HCONV_ABLATE & 1)) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          yv[k] = yv[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * yv[k]));
      }
      w[2 * hh] = ok ? pack2<DT>(yv[0], yv[1]) : 0u;
      w[2 * hh + 1] = ok ? pack2<DT>(yv[2], yv[3]) : 0u;
      if constexpr (SIDE) {
        rw[2 * hh] = pack2<DT>(xv[0], xv[1]);
        rw[2 * hh + 1] = pack2<DT>(xv[2], xv[3]);
      }
    }
    if (p < 2 || tid16 < (1296u - 1024u) * 16u)  // the item exists (pass 2 has 272 of them)
      *reinterpret_cast<uint4*>(lds + pbuf + (pkp >> 21) * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
    if (SIDE && side) {
      const unsigned voff = ((imask >> p) & 1u) ? (pkp & 0x1fffffu) * (unsigned)(C * 2) + (tid16 & 48u) : OOBA;  // + 16 q
      i32x4 r = rRaw;
      r.z = cc < C ? r.z : 0;
      const u32x4 d = {rw[0], rw[1], rw[2], rw[3]};
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen"
                   :: "v"(d), "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(cc < C ? (unsigned)(cc * 2) : 0u)) : "memory");
    }
  };

  // ---- fragment addressing ----
  // A (pixels): patch pixel (tile row R + ky, li + kx) -> byte (R + ky) * 1536 + (li + kx) * 64 + swizzled slot; the swizzle
  // bit is bit 2 of the pixel index = bit 2 of (li + kx) (the pitch is a multiple of 8): one base per kx
  unsigned abase[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int c = li + kx;
    abase[kx] = (unsigned)(wm * MT * 1536 + c * 64 + ((lg ^ (((c >> 2) & 1) << 1)) << 4));
  }
  const unsigned wfrag = (unsigned)(OFF_W + (wn * 16 * NR + li) * 64 + ((lg ^ (((li >> 2) & 1) << 1)) << 4));

  f32x4 acc[NR][MT];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one tap-step (K = 32): idx = 0..17 within the 64-channel chunk (scalar); weights at byte `wbase` (+ this half's tap-step)
  auto compute = [&](int idx, unsigned wbase) {
    const int sub = idx >= 9 ? 1 : 0, tap = idx - 9 * sub;
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const unsigned ab_ = kx == 0 ? abase[0] : (kx == 1 ? abase[1] : abase[2]);
    const char* pa = lds + (ab_ + (unsigned)(sub * PB + ky * 1536));
    const char* pw = lds + (wbase + (unsigned)((idx & 1) * WTS));
    // weights resident (NR fragments), pixel fragments streamed: the order that keeps the fewest registers live
    uint4 fb[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) fb[ni] = *reinterpret_cast<const uint4*>(pw + ni * 1024);
    if (HCONV_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const uint4 fa = *reinterpret_cast<const uint4*>(pa + mi * 1536);
      if (HCONV_ABLATE & 2) {
        asm volatile("" :: "v"(fa));
        continue;
      }
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
        // swapped: D[row = n-in-tile = 4*lg + r][col = pixel-in-row = li]
        acc[ni][mi] = HT<DT>::mfma16(fb[ni], fa, acc[ni][mi]);
    }
    if (HCONV_ABLATE & 2) asm volatile("" :: "v"(fb[0]), "v"(fb[NR - 1]));
    if (HCONV_SETPRIO) __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: affine table, patch image a of chunk 0 (three passes staged in still empty LDS areas) ----
  if (HCONV_PRO) {
    // the three passes of patch image a are staged where nothing lives yet: the patch staging area, ring slot 1 and image b;
    // ring slot 0 receives the first weight tile meanwhile.  The first step's barrier orders every later writer of those
    // areas (step 0 refills slot 1 and writes image b only after it).
    issue_w(0, 0, 0u);
    issue_patch(0, 0, (unsigned)OFF_STG);
    issue_patch(1, 0, (unsigned)(OFF_W + WSLOT));
    issue_patch(2, 0, (unsigned)OFF_PB);
    if (norm) {
      const uint4* src = reinterpret_cast<const uint4*>(g.ab + (int64_t)img * 2 * C);
      for (int i = tid; i < (2 * C) / 4; i += 512) smem[OFF_AB / 16 + i] = src[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    consume(0, 0, (unsigned)OFF_STG, (unsigned)OFF_PA);
    consume(1, 0, (unsigned)(OFF_W + WSLOT), (unsigned)OFF_PA);
    consume(2, 0, (unsigned)OFF_PB, (unsigned)OFF_PA);
    issue_patch(0, 32, (unsigned)OFF_STG);
  } else {
    if (norm) {
      const uint4* src = reinterpret_cast<const uint4*>(g.ab + (int64_t)img * 2 * C);
      for (int i = tid; i < (2 * C) / 4; i += 512) smem[OFF_AB / 16 + i] = src[i];
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) issue_patch(p, 0, (unsigned)(OFF_W + p * STG));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int p = 0; p < 3; ++p) consume(p, 0, (unsigned)(OFF_W + p * STG), (unsigned)OFF_PA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // staging areas free: the ring may be filled
    issue_w(0, 0, 0u);
    issue_patch(0, 32, (unsigned)OFF_STG);
  }

  // ---- main loop.  Step J of chunk c multiplies tap-steps 2J, 2J+1; around it:
  //   weights of the next step -> the other ring slot (waves 0-3 right after the barrier, waves 4-7 between the halves);
  //   patch passes: image b of chunk c is written in steps 0..2 (read from step 4 on), image a of chunk c + 1 in steps
  //   5..7 (image a of chunk c is last read in step 4); every pass is fetched one step before it is transformed.
  //   Queue of a wave when it reaches the start of a step: [W DMAs of this step][patch DMAs (2) if the last step issued
  //   a pass] -> vmcnt(2 | 0); when it transforms a pass: [patch DMAs][W DMAs of the next step (NWD)] -> vmcnt(NWD).
  // The loop is ROLLED (one 80-MFMA body, the step's position in the chunk is scalar state): with the nine steps
  // unrolled the register allocator renames accumulators across the body and spills them, and every scratch reload is a
  // vector-memory load hipcc waits for with vmcnt(0) — it drains the DMA queue.
  int J = 0, c = 0;
  const int nsteps = nch * 9;
  for (int st = 0; st < nsteps; ++st) {
    const bool prev_issued = !(J == 3 || J == 4 || J == 8);  // steps 2, 3, 7 issue no patch pass
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(imask), "+v"(tid16));
    if (prev_issued) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned par = (unsigned)st & 1u;  // (c + J) & 1 == (9 c + J) & 1
    const unsigned wbase = wfrag + par * (unsigned)WSLOT;
    const int jn = J == 8 ? 0 : J + 1, cn = J == 8 ? c + 1 : c;
    // (a macro, not a nested lambda: with one more closure level SROA gives up and the kernel-argument block lands in scratch)
#define HCONV_PATCH_WORK()                                                                                     \
    do {                                                                                                         \
      if (J <= 2) {                                                                                              \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((HCONV_ABLATE & 4) ? 0 : NWD) : "memory");                     \
        consume(J, c * 64 + 32, (unsigned)OFF_STG, (unsigned)OFF_PB);                                            \
      } else if (J >= 5 && J <= 7) {                                                                             \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((HCONV_ABLATE & 4) ? 0 : NWD) : "memory");                     \
        consume(J - 5, (c + 1) * 64, (unsigned)OFF_STG, (unsigned)OFF_PA);                                       \
      }                                                                                                          \
      if (J <= 1) issue_patch(J + 1, c * 64 + 32, (unsigned)OFF_STG);                                            \
      else if (J >= 4 && J <= 6) issue_patch(J - 4, (c + 1) * 64, (unsigned)OFF_STG);                            \
      else if (J == 8) issue_patch(0, (c + 1) * 64 + 32, (unsigned)OFF_STG);                                     \
    } while (0)
    if (!ts) issue_w(jn, cn, par ^ 1u);
    compute(2 * J, wbase);
    if (ts) issue_w(jn, cn, par ^ 1u);
    if (HCONV_FENCE) __builtin_amdgcn_sched_barrier(0);
    if (!HCONV_LATE_CONSUME || !ts) HCONV_PATCH_WORK();
    if (HCONV_FENCE) __builtin_amdgcn_sched_barrier(0);
    compute(2 * J + 1, wbase);
    if (HCONV_LATE_CONSUME && ts) {
      if (HCONV_FENCE) __builtin_amdgcn_sched_barrier(0);
      HCONV_PATCH_WORK();
    }
#undef HCONV_PATCH_WORK
    if (++J == 9) { J = 0; ++c; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing zero-fill DMAs must not outlive the block's LDS

  // ---- epilogue: lane holds out[pixel (Y0 + wm MT + mi, X0 + li)][n = N0 + 16 NR wn + 16 ni + 4 lg + r] ----
  {
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int64_t Mbase = ((int64_t)img * g.H + Y0) * g.Wd + X0;
    const unsigned ext = (unsigned)(((15 * g.Wd + 15) * g.N + g.N) * 4);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc((char*)g.out + Mbase * g.N * 4, 0, (int)ext, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
        g.res ? (char*)const_cast<float*>(g.res) + Mbase * g.N * 4 : nullptr, 0, g.res ? (int)ext : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bias =
        __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.N * 4 : 0, 0x00020000);
    const float* ibp = g.img_bias ? g.img_bias + (int64_t)(img / g.imgs_per_bias_row) * g.ldib : nullptr;
    const __amdgpu_buffer_rsrc_t r_imgb = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(ibp), 0, ibp ? g.N * 4 : 0, 0x00020000);
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& r, unsigned off) -> f32x4 {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    };
    const bool epi_silu = g.epi_silu != 0;
    f32x4 bv[NR];
    bool col_ok[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      const int nn = N0 + wn * 16 * NR + ni * 16 + 4 * lg;
      col_ok[ni] = nn < g.N;
      bv[ni] = ld4(r_bias, col_ok[ni] ? (unsigned)nn * 4u : OOB) + ld4(r_imgb, col_ok[ni] ? (unsigned)nn * 4u : OOB);
    }
    // ---- optional: GroupNorm statistics of the tensor this launch writes, per 256-pixel tile and channel ----
    // (mean, sum of squared deviations from that mean).  Without a residual the stored value is (acc + bias) * scale, so
    // the statistics follow from those of the accumulators, which are all still in registers here: an exact two-pass
    // computation (sum over the wave's MT rows in registers, over the 16 pixels of a row by DPP, over the WM waves through
    // LDS, all in a fixed order) BEFORE the store loop — it adds no live state to the loop below.
    // mimo_group_norm_stats_slabs merges the tiles of an image (Chan's update, in double): the norm that consumes this
    // tensor makes no statistics pass over HBM.  8x fewer partials than the 32-row slabs of mimo_conv2d_ext.
    if constexpr (TSTATS == 1) {
      float* red = reinterpret_cast<float*>(lds);  // [2][WM][BN]; the patch images are dead once every wave is past its last step
      float* red2 = red + WM * BN;
      auto dpp = [](float v, auto ctrl_c) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_c)::value, 0xf, 0xf, true));
      };
      auto row16_sum = [&](float x) {
        x += dpp(x, IC<0xB1>{});   // quad_perm [1,0,3,2]
        x += dpp(x, IC<0x4E>{});   // quad_perm [2,3,0,1]
        x += dpp(x, IC<0x141>{});  // row_half_mirror
        x += dpp(x, IC<0x140>{});  // row_mirror
        return x;
      };
      const int colw = wn * 16 * NR + 4 * lg;  // this lane's first column inside the block's BN
      auto tile_total = [&](const float* base, int ni) -> f32x4 {  // sum over the WM waves' partials, fixed order
        f32x4 t = *reinterpret_cast<const f32x4*>(base + colw + ni * 16);
#pragma unroll
        for (int w = 1; w < WM; ++w) t += *reinterpret_cast<const f32x4*>(base + w * BN + colw + ni * 16);
        return t;
      };
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is past its last fragment read
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) {  // pass 1: sums
        f32x4 t = acc[ni][0];
#pragma unroll
        for (int mi = 1; mi < MT; ++mi) t += acc[ni][mi];
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = row16_sum(t[r]);
        if (li == 0) *reinterpret_cast<f32x4*>(red + wm * BN + colw + ni * 16) = t;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) {  // pass 2: centred sums of squares
        const f32x4 mean = tile_total(red, ni) * (1.0f / 256.0f);
        f32x4 q = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const f32x4 d = acc[ni][mi] - mean;
          q += d * d;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = row16_sum(q[r]);
        if (li == 0) *reinterpret_cast<f32x4*>(red2 + wm * BN + colw + ni * 16) = q;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wm == 0 && li == 0) {
        const int64_t tile = ((int64_t)img * g.tiles_y + ty) * g.tiles_x + tx;
        float* dst = g.tstats + tile * 2 * g.N;
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) {
          const int nn = N0 + colw + ni * 16;
          if (nn < g.N) {
            *reinterpret_cast<f32x4*>(dst + nn) = (tile_total(red, ni) * (1.0f / 256.0f) + bv[ni]) * g.out_scale;
            *reinterpret_cast<f32x4*>(dst + g.N + nn) = tile_total(red2, ni) * (g.out_scale * g.out_scale);
          }
        }
      }
    }
    auto run = [&](auto has_res_c) {
      constexpr bool HAS_RES = decltype(has_res_c)::value != 0;
      if constexpr (TSTATS == 2) {
        // ---- store loop with GroupNorm statistics of the stored values (residual included), per 32-pixel slab = two image
        // rows of the tile, both owned by this wave: an exact two-pass computation on the final values of the pair (sum over
        // the two rows in registers, over the 16 pixels of a row by DPP), the layout and slab size of mimo_gemm_ext's
        // column statistics (slabs of an image are contiguous: tile * 8 + row pair) ----
        auto dpp = [](float v, auto ctrl_c) {
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_c)::value, 0xf, 0xf, true));
        };
        auto row16_sum = [&](float x) {
          x += dpp(x, IC<0xB1>{});
          x += dpp(x, IC<0x4E>{});
          x += dpp(x, IC<0x141>{});
          x += dpp(x, IC<0x140>{});
          return x;
        };
        const int64_t tile = ((int64_t)img * g.tiles_y + ty) * g.tiles_x + tx;
        float* const sdst = g.tstats + (tile * 8 + wm * (MT / 2)) * 2 * g.N + N0 + wn * 16 * NR + 4 * lg;
#pragma unroll
        for (int sp = 0; sp < MT / 2; ++sp) {
          const unsigned prow_a = (unsigned)(((wm * MT + 2 * sp) * g.Wd + li) * g.N), prow_b = prow_a + (unsigned)(g.Wd * g.N);
          f32x4 ra[NR], rb[NR];
          if (HAS_RES) {
#pragma unroll
            for (int ni = 0; ni < NR; ++ni) {
              const unsigned c = (unsigned)(N0 + wn * 16 * NR + ni * 16 + 4 * lg);
              ra[ni] = ld4(r_res, col_ok[ni] ? (prow_a + c) * 4u : OOB);
              rb[ni] = ld4(r_res, col_ok[ni] ? (prow_b + c) * 4u : OOB);
            }
          }
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) {
            f32x4 va = acc[ni][2 * sp] + bv[ni], vb = acc[ni][2 * sp + 1] + bv[ni];
            if (HAS_RES) { va += ra[ni]; vb += rb[ni]; }
            va *= g.out_scale;
            vb *= g.out_scale;
            const unsigned c = (unsigned)(N0 + wn * 16 * NR + ni * 16 + 4 * lg);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, va), r_out, col_ok[ni] ? (prow_a + c) * 4u : OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vb), r_out, col_ok[ni] ? (prow_b + c) * 4u : OOB, 0, 0);
            f32x4 t = va + vb;
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = row16_sum(t[r]);
            const f32x4 mean = t * (1.0f / 32.0f);
            const f32x4 d0 = va - mean, d1 = vb - mean;
            f32x4 q = d0 * d0 + d1 * d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = row16_sum(q[r]);
            if (li == 0 && col_ok[ni]) {
              *reinterpret_cast<f32x4*>(sdst + (int64_t)sp * 2 * g.N + ni * 16) = mean;
              *reinterpret_cast<f32x4*>(sdst + (int64_t)sp * 2 * g.N + g.N + ni * 16) = q;
            }
          }
        }
        return;
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const unsigned prow = (unsigned)(((wm * MT + mi) * g.Wd + li) * g.N);
        f32x4 rr[NR];
        if (HAS_RES) {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni)
            rr[ni] = ld4(r_res, col_ok[ni] ? (prow + (unsigned)(N0 + wn * 16 * NR + ni * 16 + 4 * lg)) * 4u : OOB);
        }
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) {
          f32x4 v = acc[ni][mi] + bv[ni];
          if (epi_silu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
          }
          if (HAS_RES) v += rr[ni];
          v *= g.out_scale;
          const unsigned off = col_ok[ni] ? (prow + (unsigned)(N0 + wn * 16 * NR + ni * 16 + 4 * lg)) * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, off, 0, 0);
        }
      }
    };
    if constexpr (TSTATS == 2) run(IC<1>{});  // (launched with a residual only)
    else if (g.res) run(IC<1>{});
    else run(IC<0>{});
  }
}

inline int cus_h() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return cus;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <int DT>
int launch_h(HArgs& g, hipStream_t st) {
  const int C = g.C1 + g.C2;
  int bn, abmax;
  // every weight row of a tile must exist (the DMAs step rows through the scalar offset, which is not range checked)
  if (g.N % 320 == 0) { bn = 320; abmax = 7680; }
  else if (g.N % 256 == 0) { bn = 256; abmax = 20480; }
  else if (g.N % 128 == 0) { bn = 128; abmax = 20480; }
  else return MIMO_EINVAL;
  if (g.ab && C * 8 > abmax) return MIMO_EINVAL;
  g.tiles_n = (g.N + bn - 1) / bn;
  const int64_t nwg = (int64_t)g.n * g.tiles_y * g.tiles_x * g.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
#define HC_LAUNCH(NR_, WM_, WN_, AB_)                                                                                                   \
  do {                                                                                                                                \
    const dim3 gr((unsigned)nwg), bl(512);                                                                                            \
    if (!g.ab && g.tstats) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, false, false, 1>), gr, bl, 0, st, g);             \
    else if (!g.ab) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, false, false, 0>), gr, bl, 0, st, g);                    \
    else if (g.raw && g.tstats) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, true, true, 1>), gr, bl, 0, st, g);          \
    else if (g.raw) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, true, true, 0>), gr, bl, 0, st, g);                      \
    else if (g.tstats && g.res) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, true, false, 2>), gr, bl, 0, st, g);         \
    else if (g.tstats) hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, true, false, 1>), gr, bl, 0, st, g);                  \
    else hipLaunchKernelGGL((hconv_kernel<DT, NR_, WM_, WN_, AB_, true, false, 0>), gr, bl, 0, st, g);                                \
  } while (0)
  if (bn == 320) HC_LAUNCH(5, 2, 4, 7680);
  else if (bn == 256) HC_LAUNCH(4, 2, 4, 20480);
  else HC_LAUNCH(4, 4, 2, 20480);
#undef HC_LAUNCH
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

}  // namespace

extern "C" int mimo_conv3x3_fused(int dtype, const float* x1, int C1, const float* x2, int C2, const float* ab, int silu,
                                  const void* W, int64_t ldw, float* out, const mimo_hconv_params* p, const float* bias,
                                  const float* img_bias, const float* residual, void* raw_out, float* tile_stats,
                                  float out_scale, unsigned flags, void* stream) {
  if (!x1 || !W || !out || !p) return MIMO_EINVAL;
  if (p->n <= 0 || p->H <= 0 || p->W <= 0 || (p->H & 15) || (p->W & 15) || p->Cout <= 0 || (p->Cout & 3)) return MIMO_EINVAL;
  if (C1 <= 0 || C2 < 0 || (C2 > 0 && !x2)) return MIMO_EINVAL;
  const int C = C1 + C2;
  if ((C & 31) || (C2 > 0 && ((C1 & 63) || (C2 & 31)))) return MIMO_EINVAL;
  if (flags & ~(MIMO_EPI_SILU | MIMO_EPI_OUT_F32 | MIMO_EPI_RES_F32)) return MIMO_EINVAL;
  if ((ab != nullptr) != (silu != 0)) return MIMO_EINVAL;  // instantiated: affine + SiLU (ResBlocks) | plain cast (Upsample)
  if (raw_out && !ab) return MIMO_EINVAL;
  // statistics: per 256-pixel tile from the accumulators ((acc + bias) * scale: no residual), or per 32-pixel slab from the
  // stored values when a residual is added (instantiated for the ResBlock's second convolution: affine, no side output)
  if (tile_stats && ((flags & MIMO_EPI_SILU) || (residual && (!ab || raw_out)))) return MIMO_EINVAL;
  if (!(flags & MIMO_EPI_OUT_F32) || (residual && !(flags & MIMO_EPI_RES_F32))) return MIMO_EINVAL;  // fp32 output / residual only
  if (p->upsample2x && ((p->H & 1) || (p->W & 1) || raw_out)) return MIMO_EINVAL;
  if (ldw < 9 * (int64_t)C || (ldw & 7)) return MIMO_EINVAL;
  if (!al16(x1) || (x2 && !al16(x2)) || !al16(W) || !al16(out) || (ab && !al16(ab)) || (residual && !al16(residual)) ||
      (raw_out && !al16(raw_out)) || (bias && !al16(bias)) || (img_bias && !al16(img_bias)) || (tile_stats && !al16(tile_stats)))
    return MIMO_EINVAL;
  const int ldib = p->img_bias_ld > 0 ? p->img_bias_ld : p->Cout;
  if (img_bias && (ldib & 3)) return MIMO_EINVAL;
  HArgs g{};
  g.x1 = x1; g.x2 = C2 > 0 ? x2 : nullptr; g.ab = ab; g.W = (const uint16_t*)W; g.out = out;
  g.bias = bias; g.img_bias = img_bias; g.res = residual; g.raw = (uint16_t*)raw_out; g.tstats = tile_stats;
  g.ldw = ldw; g.ldib = ldib; g.C1 = C1; g.C2 = C2; g.n = p->n; g.H = p->H; g.Wd = p->W;
  g.ups = p->upsample2x ? 1 : 0;
  g.Hs = g.ups ? p->H / 2 : p->H; g.Ws = g.ups ? p->W / 2 : p->W;
  g.N = p->Cout; g.tiles_x = p->W / 16; g.tiles_y = p->H / 16;
  g.silu = silu ? 1 : 0; g.epi_silu = (flags & MIMO_EPI_SILU) ? 1 : 0;
  g.imgs_per_bias_row = p->imgs_per_bias_row > 0 ? p->imgs_per_bias_row : 1;
  g.out_scale = out_scale;
  // 32-bit offsets inside one image / one tile / the weight matrix; 2 GiB keeps OOBA + soffset out of range
  const int64_t src_px = (int64_t)g.Hs * g.Ws;
  const int64_t wb = (int64_t)g.N * ldw * 2;
  // the source pixel index travels in 21 bits of pk[] (slot << 21) with Hs * Ws itself as the out-of-image sentinel
  if (src_px >= (1LL << 21)) return MIMO_EINVAL;
  if (src_px * C1 * 4 >= 0x80000000LL || src_px * (int64_t)C2 * 4 >= 0x80000000LL || (int64_t)p->H * p->W * C * 2 >= 0x80000000LL ||
      wb >= 0x80000000LL || ((int64_t)15 * p->W + 16) * g.N * 4 >= 0x80000000LL)
    return MIMO_EINVAL;
  g.w_bytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16) return launch_h<MIMO_F16>(g, st);
  if (dtype == MIMO_BF16) return launch_h<MIMO_BF16>(g, st);
  return MIMO_EDTYPE;
}
