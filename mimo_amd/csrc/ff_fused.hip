// ff_fused.hip — the tail of a transformer block as ONE kernel (gfx950), C = 320 (level 0 of the UNets).  Three entry points
// share the kernel (template MODE):
//
//   0  mimo_ff_fused         out[M, C] = res + GEGLU(A @ W1^T + b1) @ W2^T + b2                 (A = LayerNorm output; out half)
//   1  mimo_ff_proj_fused    out32     = x + (res + FF(A)) @ Wp^T + bp                          (+ the owning transformer's proj_out)
//   2  mimo_block_tail_fused y = res + O @ Wo^T + bo (+ img_bias);  n = LayerNorm(y);  out32 = x + (y + FF(n)) @ Wp^T + bp
//                            (+ the attention's to_out, its residual, the collapsed cross-attention vector and the LayerNorm)
//
//   3  mimo_block_head_fused  y = (res +) A @ Wi^T + bi;  n = LayerNorm(y) (+ pe[frame]);  qkv = n @ Wqkv^T   (A half)
//   4  (same entry point)     A = half(x * ga[img] + gb[img]): the GroupNorm in front of proj_in applied while the fp32
//                             block input is loaded (round 5: everything a transformer block does BEFORE its attention core)
//
// The feed-forward core, which the other modes wrap with projection steps on the same weight stream (see MODE below):
//
// replaces diffusers FeedForward(GEGLU, Linear) + the residual add of src/models/attention.py:428-429 /
// motion_module.py:258 as two launches (mimo_gemm GEGLU, then mimo_gemm + residual) with the [M, 4C] intermediate
// (503 MB at 512x512x24f) written to and re-read from HBM.  Here the intermediate never leaves the chip: a block owns a
// 128-row panel; per 32-column chunk of the hidden dimension it computes the chunk (FF1 + GEGLU) and immediately
// multiplies it into the FF2 accumulators, which are initialised with the residual and stay in registers for the
// whole panel.
//
// Block = 8 waves.  Waves w and w + 4 (the two waves of one SIMD) own the SAME 32 rows and split the columns:
//   FF1: wave half s computes value/gate tile pair s of the 64-row W1 tile (16 hidden columns) -> 4 values per lane and row
//        tile; the two halves swap their values lane-to-lane through 1 KB of LDS (every lane needs exactly what its twin
//        lane in the partner wave holds: MFMA results have lanes along rows, 4 consecutive columns per lane),
//   FF2: wave half s accumulates output columns [160 s, 160 s + 160) for its 32 rows (80 fp32 VGPRs).
// The hidden chunk becomes the B operand of the FF2 MFMA straight from registers: the k order inside a 32-wide chunk is
// whatever the accumulator layout gives (lane group g holds columns 4g..4g+3 and 16+4g..16+4g+3), and W2 is packed with
// the same permutation of its K axis (mimo_amd.packing.pack_ff2_kperm), so no shuffle is needed.
// W stream (as gemm_stream.hip): per step one 64 x 320 tile of W1 (40 KB) and one 320 x 32 slice of W2 (20 KB) by LDS-DMA
// into 2-deep rings, swizzled on the source address; the stream is continuous across panels.
// Pipeline: ONE barrier per step.  FF2 of chunk c runs in step c + 1 (its W2 slice is streamed one step behind the W1
// tile, the partner's half of the chunk was written before the barrier).  Only the half-1 waves issue the DMAs, which
// shifts them behind their SIMD partners: one wave's GELU (VALU) and DMA issue sit under the other's MFMAs.
#include "common.hip.h"

#include "ff_fused.hip.h"

#ifndef FF_HEAD_RING   // block head: depth of the W tile ring (2: the feed-forward modes' stage layout; 3: three 40 KB tiles, two ahead).
#define FF_HEAD_RING 2 // Measured equal (round 6, profiles/r6_block_head_ring_depth_ab.txt: 0.294-0.300 ms either way, forward 68.4-68.6): the
#endif                 // step's wait is not the tile's latency; 2 is shipped, 3 stays as a build variant (-DFF_HEAD_RING=3)

namespace {

// MODE 0: feed-forward only (half output); 1: + the block's output projection; 2: + the attention output projection and
// the LayerNorm in front (the whole tail of a transformer block after its attention core)
template <int DT, int MODE>
__global__ __launch_bounds__(512, 2) void ff_fused_kernel(const FFArgs g) {
  constexpr bool HEAD = MODE >= 3;                      // block head: projection -> LayerNorm -> QKV, no feed-forward
  constexpr bool TAIL = MODE == 1 || MODE == 2;
  constexpr int NPRE = MODE >= 2 ? NTAIL : 0;           // W tiles of the folded leading projection (to_out | proj_in)
  constexpr int NFF = HEAD ? 0 : NSTEP;                 // feed-forward positions
  constexpr int NQKV = 3 * C / 64;                      // block head: W tiles of the fused QKV projection
  constexpr int NPOS = NPRE + NFF + (TAIL ? NTAIL : 0) + (HEAD ? NQKV : 0);   // stream positions (W tiles of the W1 region) per panel
  // W ring.  The feed-forward modes: 2 stages of [W1 tile | W2 slice] (60 KB each).  Block head with FF_HEAD_RING = 3 (build
  // variant): no W2 slices exist, so the same 120 KB hold THREE 40 KB tiles and the stream runs TWO positions ahead — a QKV
  // step's counted wait then asks for a tile issued two steps earlier.  It measured equal to the 2-deep ring: the ~1 300 of a
  // step's ~3 800 cycles spent in wait + barrier (profiles/r5_block_head_phase_trace.txt) are not the tile's latency
  constexpr int RING = HEAD ? FF_HEAD_RING : 2;
  constexpr unsigned SLOT = (HEAD && FF_HEAD_RING == 3) ? (unsigned)W1_TILE : (unsigned)STAGE;
  static_assert(RING * SLOT <= (unsigned)BIAS_OFF, "the ring must end below the bias image");
  __shared__ __attribute__((aligned(16))) uint4 smem[LDS_BYTES / 16];  // ONE LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned pr = wave_u & 3u, sh = wave_u >> 2;   // row group (32 rows) and column half
  const int lg = lane >> 4, li = lane & 15;
  const unsigned npanels = (unsigned)((g.M + BM - 1) / BM);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  float* const bias_lds = reinterpret_cast<float*>(reinterpret_cast<char*>(&smem[0]) + BIAS_OFF);
  for (int n = tid; n < 8 * C; n += 512) bias_lds[n] = g.b1 ? g.b1[n] : 0.f;
  for (int n = tid; n < C; n += 512) bias_lds[8 * C + n] = g.b2 ? g.b2[n] : 0.f;
  for (int n = tid; n < C; n += 512) bias_lds[9 * C + n] = (TAIL && g.bp) ? g.bp[n] : 0.f;
  for (int n = tid; n < C; n += 512) {
    bias_lds[10 * C + n] = (MODE >= 2 && g.bo) ? g.bo[n] : 0.f;
    bias_lds[11 * C + n] = MODE >= 2 ? g.ln_gamma[n] : 0.f;
    bias_lds[12 * C + n] = MODE >= 2 ? g.ln_beta[n] : 0.f;
  }

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const i32x4 rW1 = make_rsrc(g.W1, (unsigned)(MODE >= 2 ? NPOS * 64 : 8 * C) * (unsigned)ROWB1);
  const i32x4 rW2 = make_rsrc(g.W2, (unsigned)C * (unsigned)(HID * 2));
  const i32x4 rWp = make_rsrc(MODE == 1 ? g.Wp : g.W1, MODE == 1 ? (unsigned)C * (unsigned)ROWB1 : 0u);
  constexpr unsigned OOBA = 0x80000000u;

  // ---- W stream.  Pieces of 1 KB per step: 0..39 = the W1 tile, 40..59 = the W2 slice; wave w moves pieces w, w + 8, ...
  // W1 image: five K-blocks of [64 rows x 128 B]; chunk c of a row's 128-byte segment stored at c ^ (row & 7).
  // W2 image: row n (output column) at n * 64 B, chunk q (8 permuted k) stored at q ^ (2 * ((n >> 3) & 1)): conflict-free
  // for the ds_read_b128 lane groups (brute-forced against the grouping of MI355X_MICROARCH.md). ----
  // Everything a lane contributes to a DMA address is loop-invariant (one VGPR per operand); the piece index and the
  // stream position travel in the scalar offset:
  //   W1 piece p = (K-block kb = p / 8, row group rg = p % 8): 8 rows x 128 B; lane l fetches row 8 rg + (l >> 3), logical
  //   16-byte chunk (l & 7) ^ (row & 7) of the K-block -> byte (8 rg + (l >> 3)) * 640 + kb * 128 + chunk * 16 of the tile
  //   W2 piece d: 16 rows x 64 B; lane l fetches row 16 d + (l >> 2), logical chunk (l & 3) ^ (2 * ((l >> 5) & 1))
  const unsigned w1_lane = ((unsigned)lane >> 3) * (unsigned)ROWB1 + ((((unsigned)lane & 7u) ^ (((unsigned)lane >> 3) & 7u)) << 4);
  const unsigned w2_lane = ((unsigned)lane >> 2) * (unsigned)(HID * 2) + ((((unsigned)lane & 3u) ^ (2u * (((unsigned)lane >> 5) & 1u))) << 4);
  auto dma = [&](const i32x4& r_, unsigned voff, unsigned soff, unsigned lds_dst) {
    // (every scalar operand said to be scalar: under register pressure the compiler otherwise parks descriptors in VGPRs)
    const i32x4 r = {__builtin_amdgcn_readfirstlane(r_.x), __builtin_amdgcn_readfirstlane(r_.y),
                     __builtin_amdgcn_readfirstlane(r_.z), __builtin_amdgcn_readfirstlane(r_.w)};
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory", "m0");
  };
  const unsigned my_panels = blockIdx.x < npanels ? (npanels - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
  const unsigned total = my_panels * (unsigned)NPOS;
  unsigned ld_t = 0, ld_j = 0;
  // issue(): the W1 tile of stream position ld_t (hidden chunk ld_j) into W1 stage ld_t & 1, and the W2 slice of position
  // ld_t - 1 into W2 stage (ld_t - 1) & 1 (FF2 of a chunk runs one step after its FF1).  Only the waves of column half 1
  // issue (15 pieces each): their ~1000 cycles of DMA issue shift them behind their SIMD partners of half 0, which start
  // the step's MFMAs at once — the two waves of a SIMD then alternate between the matrix pipe and VALU / issue work
  // with ONE code path (a second, reordered path for half 0 cost 15 scratch reloads per step at the 256-register cap).
  auto issue_next = [&]() {
    if (sh == 1u && !(FF_ABLATE(g, 1) && ld_t >= 2u)) {
      const bool live1 = ld_t < total;
      const unsigned j2 = ld_j == 0u ? (unsigned)NPOS - 1u : ld_j - 1u;      // position ld_t - 1 inside its panel
      // (only the feed-forward positions have a W2 slice)
      const bool live2 = NFF > 0 && ld_t >= 1u && ld_t <= total && j2 >= (unsigned)NPRE && j2 < (unsigned)(NPRE + NFF);
      const unsigned dst1 = smem_base + (ld_t % (unsigned)RING) * SLOT;
      const unsigned dst2 = smem_base + ((ld_t + 1u) & 1u) * (unsigned)STAGE + (unsigned)W1_TILE;
      // MODE 2: the W1 region's tiles are one stream in position order; MODE 1: the projection tiles are a second tensor
      const bool tail_tile = MODE == 1 && ld_j >= (unsigned)NSTEP;
      const i32x4 r1 = tail_tile ? rWp : rW1;
      const unsigned tile = tail_tile ? ld_j - (unsigned)NSTEP : ld_j;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const unsigned p = pr + 4u * i, kb = p >> 3, rg = p & 7u;
        dma(r1, live1 ? w1_lane : OOBA, tile * (unsigned)W1_TILE + rg * (8u * ROWB1) + kb * 128u, dst1 + p * 1024u);
      }
      if (live2) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const unsigned d = pr + 4u * i;
          dma(rW2, w2_lane, (j2 - (unsigned)NPRE) * 64u + d * (16u * HID * 2u), dst2 + d * 1024u);
        }
      }
    }
    // (wave-uniform by construction; said explicitly, the descriptors and LDS targets derived from them must stay scalar)
    ld_t = __builtin_amdgcn_readfirstlane(ld_t + 1u);
    ld_j = __builtin_amdgcn_readfirstlane(ld_j + 1u == (unsigned)NPOS ? 0u : ld_j + 1u);
  };

  // W1 fragment: tile row ni * 16 + li, logical chunk 4 ks + lg; this wave's tiles are ni = 2 sh (value), 2 sh + 1 (gate)
  const unsigned bq0 = (unsigned)li * 8u + (unsigned)(lg ^ (li & 7));          // uint4 index inside a K-block, even k-steps
  const unsigned bq1 = (unsigned)li * 8u + (unsigned)((4 + lg) ^ (li & 7));    // odd k-steps
  // W2 fragment: row 160 sh + 16 nt + li, chunk lg ^ swizzle(li)
  const unsigned w2q = (unsigned)(W1_TILE / 16) + (160u * sh + (unsigned)li) * 4u + (unsigned)(lg ^ (2 * ((li >> 3) & 1)));
  const unsigned xch_mine = (unsigned)(XCH_OFF / 16) + (pr * 2u + sh) * 64u + (unsigned)lane;          // uint4 index into smem
  const unsigned xch_peer = (unsigned)(XCH_OFF / 16) + (pr * 2u + (1u - sh)) * 64u + (unsigned)lane;
  constexpr unsigned XCH_BUF = 8 * 1024 / 16;  // uint4 units between the two exchange buffers
  constexpr unsigned BIAS_Q = BIAS_OFF / 16;   // uint4 index of the bias image (4 floats per uint4)

  issue_next();     // W1 tile 0 (no W2 slice yet)
  if constexpr (RING == 3) issue_next();   // block head: the stream runs two positions ahead
  __syncthreads();  // bias image complete (the compiler drains its own loads; the DMAs are invisible to it)

  unsigned t = 0;
  [[maybe_unused]] unsigned tr = 0;
  for (unsigned panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const int64_t M0 = (int64_t)panel * BM;
    const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<uint16_t*>(g.A + M0 * g.lda), 0, (int)(((rows_valid - 1) * g.lda + C) * 2), 0x00020000);
    // (a block head may have no residual: zero records, every load returns 0)
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(g.res ? g.res + M0 * g.ldr : nullptr), 0, g.res ? (int)(((rows_valid - 1) * g.ldr + C) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.out + M0 * g.ldo), 0, (int)(((rows_valid - 1) * g.ldo + C) * 2), 0x00020000);
    // ---- this row group's 32 x 320 slice of A in MFMA operand layout (both column halves hold it) ----
    uint4 fa[2][KS];
    const unsigned a_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.lda + lg * 8) * 2));
    const unsigned a_mi = (unsigned)(16 * g.lda * 2);
    auto load_a = [&]() {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          fa[mi][ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_off + ks * 64, mi * a_mi, MIMO_LD_AUX));
    };
    // Block head, per panel: LDS copies of what depends on the image / frame a row lies in.  A 128-row panel holds rows of at
    // most TWO images / frames (rows_per_img, ln_rows_per_frame >= 128; they need not divide the panel — the reference's default
    // 784 x 784 gives 9604 rows per image): slot set 0 for rows below the boundary, set 1 from it on.  Unused b1 slots of the
    // bias image: [0, C) table row of frame 0, [C, 3C) a | b of image 0, [3C, 5C) a | b of image 1, [5C, 6C) table row of frame 1.
    // Readers of the previous panel's copies are at least one of its QKV barriers behind.
    [[maybe_unused]] int next_img = BM, next_frm = BM;   // first panel row of the second image / frame (BM: none)
    if constexpr (MODE == 4) {
      // the GroupNorm in front of proj_in, folded to x * a + b per (image, channel), is applied while the operand is loaded
      const int64_t img = M0 / g.rows_per_img;
      const int64_t nimg = (g.M + g.rows_per_img - 1) / g.rows_per_img;
      next_img = (int)((img + 1) * g.rows_per_img - M0 < BM ? (img + 1) * g.rows_per_img - M0 : BM);
      if (tid < 4 * C / 4) {   // a | b of image img, then of image img + 1 (beyond the table: zeros, rows past M)
        const __amdgpu_buffer_rsrc_t rAB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)const_cast<float*>(g.gn_ab + img * 2 * C), 0, (int)((nimg - img < 2 ? nimg - img : 2) * 2 * C * 4), 0x00020000);
        smem[BIAS_Q + (unsigned)(C / 4) + (unsigned)tid] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rAB, (unsigned)tid * 16u, 0, 0));
      }
    }
    if constexpr (HEAD) {
      if (g.ln_pe) {
        const int64_t f0 = M0 / g.ln_rows_per_frame;
        next_frm = (int)((f0 + 1) * g.ln_rows_per_frame - M0 < BM ? (f0 + 1) * g.ln_rows_per_frame - M0 : BM);
        if (tid < 2 * C / 4) {
          const int second = tid >= C / 4;
          const int64_t frame = (f0 + second) % g.ln_pe_frames;
          const __amdgpu_buffer_rsrc_t rPE = __builtin_amdgcn_make_buffer_rsrc(
              (void*)const_cast<float*>(g.ln_pe + frame * C), 0, C * 4, 0x00020000);
          smem[BIAS_Q + (unsigned)(second ? 5 * C / 4 : 0) + (unsigned)(tid - second * (C / 4))] =
              __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rPE, (unsigned)(tid - second * (C / 4)) * 16u, 0, 0));
        }
      }
    }
    if constexpr (MODE == 4) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const __amdgpu_buffer_rsrc_t rX4 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)const_cast<float*>(g.x + M0 * g.ldx), 0, (int)(((rows_valid - 1) * g.ldx + C) * 4), 0x00020000);
      const unsigned x4_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.ldx + lg * 8) * 4));
      const unsigned x4_mi = (unsigned)(16 * g.ldx * 4);
      // this lane's rows: pr * 32 + 16 mi + li; the affine of the second image starts 2C floats further
      const unsigned abq0 = BIAS_Q + (unsigned)(C / 4) + 2u * (unsigned)lg;
      const unsigned abq_mi[2] = {pinned(abq0 + ((int)(pr * 32) + li >= next_img ? (unsigned)(2 * C / 4) : 0u)),
                                  pinned(abq0 + ((int)(pr * 32) + 16 + li >= next_img ? (unsigned)(2 * C / 4) : 0u))};
      // four batches of 5 k-steps (10 loads of 16 B per lane), the next batch in flight while one is converted: all 40 loads
      // at once would need 160 registers (the compiler hoists them and spills)
      // (macros, not lambdas: arrays captured by reference two levels deep are not promoted to registers)
      f32x4 xb0[KS], xb1[KS];
#define FF_LOAD_BATCH(buf, mi, k0)                                                                                              \
  _Pragma("unroll") for (int k = 0; k < KS / 2; ++k) {                                                                          \
    buf[2 * k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX4, x4_off + ((k0) + k) * 128, (mi) * x4_mi, MIMO_LD_AUX));          \
    buf[2 * k + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX4, x4_off + ((k0) + k) * 128 + 16, (mi) * x4_mi, MIMO_LD_AUX)); \
  }                                                                                                                             \
  __builtin_amdgcn_sched_barrier(0)
#define FF_CONVERT_BATCH(buf, mi, k0)                                                                                           \
  _Pragma("unroll") for (int k = 0; k < KS / 2; ++k) {                                                                          \
    const f32x4 a0 = __builtin_bit_cast(f32x4, smem[abq_mi[mi] + (unsigned)(8 * ((k0) + k))]);                                          \
    const f32x4 a1 = __builtin_bit_cast(f32x4, smem[abq_mi[mi] + (unsigned)(8 * ((k0) + k) + 1)]);                                      \
    const f32x4 b0 = __builtin_bit_cast(f32x4, smem[abq_mi[mi] + (unsigned)(C / 4 + 8 * ((k0) + k))]);                                  \
    const f32x4 b1 = __builtin_bit_cast(f32x4, smem[abq_mi[mi] + (unsigned)(C / 4 + 8 * ((k0) + k) + 1)]);                              \
    const f32x4 y0 = __builtin_elementwise_fma(buf[2 * k], a0, b0), y1 = __builtin_elementwise_fma(buf[2 * k + 1], a1, b1);      \
    uint32_t w0 = pack2<DT>(y0[0], y0[1]), w1 = pack2<DT>(y0[2], y0[3]), w2 = pack2<DT>(y1[0], y1[1]), w3 = pack2<DT>(y1[2], y1[3]); \
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)); /* materialise HERE: the optimiser otherwise sinks the conversion to the first use and keeps x, a, b live */ \
    fa[mi][(k0) + k] = make_uint4(w0, w1, w2, w3);                                                                              \
  }                                                                                                                             \
  __builtin_amdgcn_sched_barrier(0)
      FF_LOAD_BATCH(xb0, 0, 0);
      FF_LOAD_BATCH(xb1, 0, KS / 2);
      FF_CONVERT_BATCH(xb0, 0, 0);
      FF_LOAD_BATCH(xb0, 1, 0);
      FF_CONVERT_BATCH(xb1, 0, KS / 2);
      FF_LOAD_BATCH(xb1, 1, KS / 2);
      FF_CONVERT_BATCH(xb0, 1, 0);
      FF_CONVERT_BATCH(xb1, 1, KS / 2);
#undef FF_LOAD_BATCH
#undef FF_CONVERT_BATCH
    } else if constexpr (MODE != 2) {
      load_a();   // (MODE 2 loads it after the per-image vector: 80 registers fewer in flight)
    }
    // ---- FF2 accumulators for columns [160 sh, 160 sh + 160): lane (li, lg) owns 4 consecutive columns of row li of each
    // 16-row tile.  MODE < 2: residual + b2.  MODE = 2: first the attention output projection accumulates on residual + bo
    // (+ the per-image vector), see below ----
    f32x4 acc2[10][2];
    const unsigned r_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.ldr + 160 * sh + 4 * lg) * 4));
    const unsigned bcol = pinned(BIAS_Q + 40u * sh + (unsigned)lg);  // this lane's column group inside a C-wide vector of the bias image
    const unsigned r_mi = (unsigned)(16 * g.ldr * 4);
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
      const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)((MODE >= 2 ? 10 : 8) * C / 4 + 4 * nt)]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc2[nt][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, r_off + nt * 64, mi * r_mi, MIMO_LD_AUX)) + bv;
    }
    // one 64-row tile of a [C, C] weight in the W1 region of stage t & 1 (rows in the tile order of pack_proj_tail: this
    // wave's output columns 32 q .. 32 q + 31 of its 160 are the tile's n-tiles 2 sh, 2 sh + 1) times the operand in fa
    auto proj = [&](auto q_c) {
      constexpr int q = decltype(q_c)::value;
      const unsigned sq = (t % (unsigned)RING) * (SLOT / 16u);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned qq = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 512);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const uint4 wf = smem[qq + (2u * sh + (unsigned)ni) * 128u];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc2[2 * q + ni][mi] = HT<DT>::mfma16(wf, fa[mi][ks], acc2[2 * q + ni][mi]);
        }
      }
    };
    // the partner wave's five k-steps of a [32 x 320] operand that both waves hold half of in accumulator layout come
    // through the exchange buffers, one k-step per round; own(kk, mi) = this wave's k-step kk (0..4) as packed halfs
    auto exchange_operand = [&](auto&& own) {
#pragma unroll
      for (int kk = 0; kk < 5; ++kk) {
        const uint4 m0 = own(kk, 0), m1 = own(kk, 1);
        smem[xch_mine] = m0;
        smem[xch_mine + XCH_BUF] = m1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const uint4 p0 = smem[xch_peer], p1 = smem[xch_peer + XCH_BUF];
        if (sh == 0) {
          fa[0][kk] = m0; fa[1][kk] = m1; fa[0][5 + kk] = p0; fa[1][5 + kk] = p1;
        } else {
          fa[0][kk] = p0; fa[1][kk] = p1; fa[0][5 + kk] = m0; fa[1][5 + kk] = m1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // both halves have read before the next round writes
      }
    };
    // y (in acc2) -> n = LayerNorm(y) * gamma + beta (+ pe[frame]) as the next MFMA operand in fa (MODE 2: the feed-forward's,
    // block head: the QKV projection's), straight from the accumulators
    auto ln_operand = [&]() {
      // LayerNorm over the row's 320 columns: this wave holds 160 of them, 40 per lane.  Local sum and local centred sum of
      // squares, combined with the partner's by the pairwise update (mean = (s0 + s1) / 320, M2 = q0 + q1 + 80 (m0 - m1)^2);
      // both waves evaluate the combination with the halves in the same order: bit-identical statistics.
      float rs[2], rq[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        float sum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 10; ++nt) sum += (acc2[nt][mi][0] + acc2[nt][mi][1]) + (acc2[nt][mi][2] + acc2[nt][mi][3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float ml = sum * (1.f / 160.f);
        float qq = 0.f;
#pragma unroll
        for (int nt = 0; nt < 10; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float d = acc2[nt][mi][r] - ml;
            qq = fmaf(d, d, qq);
          }
        qq += __shfl_xor(qq, 16, 64);
        qq += __shfl_xor(qq, 32, 64);
        rs[mi] = sum; rq[mi] = qq;
      }
      smem[xch_mine] = make_uint4(__float_as_uint(rs[0]), __float_as_uint(rq[0]), __float_as_uint(rs[1]), __float_as_uint(rq[1]));
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const uint4 pst = smem[xch_peer];
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // read before the operand rounds rewrite the buffer
      FF_TRACE(g, tr, 17);
      float mean[2], rstd[2];
      {
        const float ps[2] = {__uint_as_float(pst.x), __uint_as_float(pst.z)}, pq[2] = {__uint_as_float(pst.y), __uint_as_float(pst.w)};
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const float s0 = sh == 0 ? rs[mi] : ps[mi], s1 = sh == 0 ? ps[mi] : rs[mi];
          const float q0 = sh == 0 ? rq[mi] : pq[mi], q1 = sh == 0 ? pq[mi] : rq[mi];
          const float dm = (s0 - s1) * (1.f / 160.f);
          mean[mi] = (s0 + s1) * (1.f / 320.f);
          rstd[mi] = rsqrtf(((q0 + q1) + 80.f * dm * dm) * (1.f / 320.f) + g.ln_eps);
        }
      }
      // n = LayerNorm(y) * gamma + beta as the feed-forward's MFMA operand, straight from the accumulators (W1 carries the
      // K permutation of the accumulator layout)
      exchange_operand([&](int kk, int mi) -> uint4 {
        uint32_t w[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned cq = (unsigned)(4 * (2 * kk + h));
          const f32x4 gm = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(11 * C / 4) + cq]);
          f32x4 bt = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(12 * C / 4) + cq]);
          if constexpr (HEAD)  // + pe[frame of this row] (zeros without a table)
            bt += __builtin_bit_cast(f32x4, smem[bcol + cq + ((int)(pr * 32) + 16 * mi + li >= next_frm ? (unsigned)(5 * C / 4) : 0u)]);
          const f32x4 v = (acc2[2 * kk + h][mi] - mean[mi]) * rstd[mi] * gm + bt;
          w[2 * h] = pack2<DT>(v[0], v[1]);
          w[2 * h + 1] = pack2<DT>(v[2], v[3]);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
      });
    };
    if constexpr (HEAD) {
      // ---- block head: y = (res +) A @ Wi^T + bi -> out32;  qkv = LayerNorm(y) @ Wqkv^T -> out (half, 3C columns) ----
      FF_TRACE(g, tr, 10);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<0>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<1>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<2>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<3>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<4>{}); ++t;
      FF_TRACE(g, tr, 16);
      const int row0 = (int)pr * 32 + li;
      {  // the fp32 stream leaves the chip once (the attention's residual)
        const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(g.out32 + M0 * g.ldo32), 0, (int)(((rows_valid - 1) * g.ldo32 + C) * 4), 0x00020000);
        const unsigned o_off = pinned(((unsigned)row0 * (unsigned)g.ldo32 + 160u * sh + 4u * (unsigned)lg) * 4u);
        const unsigned o_mi = 16u * (unsigned)g.ldo32 * 4u;
#pragma unroll
        for (int nt = 0; nt < 10; ++nt)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc2[nt][mi]), rT, o_off + 64u * (unsigned)nt, (unsigned)mi * o_mi, MIMO_ST_AUX);
      }
      ln_operand();
      FF_TRACE(g, tr, 18);
      // 15 tiles of Wqkv (natural row order: tile q = output columns 64 q .. 64 q + 63, this wave's n-tiles 2 sh, 2 sh + 1 =
      // columns 64 q + 32 sh + [0, 32)); results leave as half in 16-byte stores (lane exchange of the paired epilogue)
      const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out + M0 * g.ldo), 0, (int)(((rows_valid - 1) * g.ldo + 3 * C) * 2), 0x00020000);
      const unsigned q_off = pinned((((unsigned)row0 + 16u * (unsigned)(lg & 1)) * (unsigned)g.ldo + 32u * sh + 4u * (unsigned)(lg & ~1)) * 2u);
      // One QKV tile.  The wait is COUNTED: vmcnt is one in-order counter for loads and stores, and a vmcnt(0) here would
      // also wait for the acknowledgement of the previous step's output stores (a memory round trip per step).  NEWER =
      // the number of this wave's vector-memory instructions issued after the DMAs of the tile about to be read: 20 y stores
      // before the first tile, 2 qkv stores afterwards (the waves of column half 0 issue no DMAs: they wait for nothing).
      auto qkv_step = [&](int q, auto newer_c) {
        constexpr int NEWER = decltype(newer_c)::value;
        FF_TRACE(g, tr, 1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NEWER) : "memory");
        FF_TRACE(g, tr, 2);
        asm volatile("s_barrier" ::: "memory");
        FF_TRACE(g, tr, 3);
        issue_next();
        FF_TRACE(g, tr, 4);
        const unsigned sq = (t % (unsigned)RING) * (SLOT / 16u);
        f32x4 a1[2][2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) a1[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!FF_ABLATE(g, 2)) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const unsigned qq = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 512);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const uint4 wf = smem[qq + (2u * sh + (unsigned)ni) * 128u];
#pragma unroll
              for (int mi = 0; mi < 2; ++mi) a1[ni][mi] = HT<DT>::mfma16(wf, fa[mi][ks], a1[ni][mi]);
            }
          }
        }
        FF_TRACE(g, tr, 5);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f32x4 va = a1[ni][0], vb = a1[ni][1];
          const auto sx = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[0], va[1]), pack2<DT>(vb[0], vb[1]), false, false);
          const auto sy = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[2], va[3]), pack2<DT>(vb[2], vb[3]), false, false);
          const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
          __builtin_amdgcn_raw_buffer_store_b128(o, rQ, q_off + 32u * (unsigned)ni, (unsigned)q * 128u, MIMO_ST_AUX);
        }
        ++t;
        FF_TRACE(g, tr, 6);
      };
      if constexpr (RING == 3) {
        // NEWER with the stream two positions ahead: behind the DMAs of the tile a step reads this wave has issued — step 0: the
        // next tile's 10 pieces + the 20 y stores (the tile itself was waited for by the last projection step's vmcnt(0));
        // step 1: the y stores + step 0's 10 pieces and 2 stores; from step 2 on: 2 stores + 10 pieces + 2 stores
        qkv_step(0, ICf<30>{});
        qkv_step(1, ICf<32>{});
#pragma unroll 1
        for (int q = 2; q < NQKV; ++q) qkv_step(q, ICf<14>{});
      } else {
        qkv_step(0, ICf<20>{});
#pragma unroll 1
        for (int q = 1; q < NQKV; ++q) qkv_step(q, ICf<2>{});
      }
      FF_TRACE(g, tr, 22);
      continue;
    }
    if constexpr (MODE == 2) {
      FF_TRACE(g, tr, 10);
      if (g.img_bias) {
        // the per-image vector (the collapsed cross-attention of the spatial blocks): the same 40 values per lane for every
        // row of an image; rows_per_img >= 128, so a panel holds rows of at most two images
        const int64_t nimg = (g.M + g.rows_per_img - 1) / g.rows_per_img;
        const __amdgpu_buffer_rsrc_t rIB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)const_cast<float*>(g.img_bias), 0, (int)(((nimg - 1) * g.ldib + C) * 4), 0x00020000);
        const int64_t img0 = M0 / g.rows_per_img;                                   // (scalar, once per panel)
        const int next0 = (int)((img0 + 1) * g.rows_per_img - M0);                  // panel row where the next image starts
        const unsigned ib_off = (unsigned)((img0 * g.ldib + 160 * sh + 4 * lg) * 4);
        if (next0 >= BM) {
#pragma unroll
          for (int nt = 0; nt < 10; ++nt) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIB, ib_off + nt * 64, 0, 0));
            acc2[nt][0] += v;
            acc2[nt][1] += v;
          }
        } else {
          const unsigned step = (unsigned)(g.ldib * 4);   // (rows past M read a vector past the table: zeros)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const unsigned o = ib_off + ((int)(pr * 32) + 16 * mi + li >= next0 ? step : 0u);
#pragma unroll
            for (int nt = 0; nt < 10; ++nt)
              acc2[nt][mi] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIB, o + nt * 64, 0, 0));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_a();
      FF_TRACE(g, tr, 11);
      // y = residual + o @ Wo^T + bo: five tiles of Wo (positions 0..4 of the panel); every tile: wait, barrier, issue the
      // next position, multiply
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<0>{}); ++t; FF_TRACE(g, tr, 12);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<1>{}); ++t; FF_TRACE(g, tr, 13);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<2>{}); ++t; FF_TRACE(g, tr, 14);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<3>{}); ++t; FF_TRACE(g, tr, 15);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<4>{}); ++t; FF_TRACE(g, tr, 16);
      ln_operand();
      FF_TRACE(g, tr, 18);
      // the feed-forward accumulates on y + b2
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) {
        const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(8 * C / 4 + 4 * nt)]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc2[nt][mi] += bv;
      }
    }

    u32x2 hm_prev[2] = {{0u, 0u}, {0u, 0u}};  // this wave's half of the previous chunk (packed), kept for its FF2
    // FF1 + GEGLU of chunk j (stream position t): leaves this wave's half in hm_prev and in the exchange buffer t & 1
    auto ff1 = [&](int j) {
      const unsigned sq = (t & 1u) * (unsigned)(STAGE / 16);
      f32x4 acc1[2][2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc1[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
      uint4 fb[KS][2];
      auto ldb = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        const unsigned q = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 512);  // K-block = 64 rows x 128 B = 512 uint4
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ks][ni] = smem[q + (2u * sh + (unsigned)ni) * 128u];  // 16 rows x 8 uint4
      };
      auto kstep = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks + 1 < KS) ldb(ICf<ks + 1>{});
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc1[ni][mi] = HT<DT>::mfma16(fb[ks][ni], fa[mi][ks], acc1[ni][mi]);
      };
      ldb(ICf<0>{});
      static_assert(KS == 10, "k-steps are spelled out");
      kstep(ICf<0>{}); kstep(ICf<1>{}); kstep(ICf<2>{}); kstep(ICf<3>{}); kstep(ICf<4>{});
      kstep(ICf<5>{}); kstep(ICf<6>{}); kstep(ICf<7>{}); kstep(ICf<8>{}); kstep(ICf<9>{});
      // GEGLU: hidden columns 32 j + 16 sh + 4 lg + r of rows 16 mi + li
      const f32x4 bval = __builtin_bit_cast(f32x4, smem[BIAS_Q + (unsigned)(16 * j) + 8u * sh + (unsigned)lg]);
      const f32x4 bgate = __builtin_bit_cast(f32x4, smem[BIAS_Q + (unsigned)(16 * j) + 8u * sh + 4u + (unsigned)lg]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const f32x4 v = acc1[0][mi] + bval, gt = acc1[1][mi] + bgate;
        if (FF_ABLATE(g, 3)) {
          hm_prev[mi].x = pack2<DT>(v[0] * gt[0], v[1] * gt[1]);
          hm_prev[mi].y = pack2<DT>(v[2] * gt[2], v[3] * gt[3]);
        } else {
          const f32x4 h = v * gelu_erf_4(gt);
          hm_prev[mi].x = pack2<DT>(h[0], h[1]);
          hm_prev[mi].y = pack2<DT>(h[2], h[3]);
        }
      }
      smem[xch_mine + (t & 1u) * XCH_BUF] = make_uint4(hm_prev[0].x, hm_prev[0].y, hm_prev[1].x, hm_prev[1].y);
    };
    // FF2 of the chunk whose stream position had parity `par`: this wave's half `hm`, the partner's from the exchange
    // buffer `par`, the W2 slice in W2 stage `par`
    auto ff2 = [&](unsigned par, const u32x2 (&hm)[2]) {
      const uint4 pe = smem[xch_peer + par * XCH_BUF];
      uint4 hf[2];  // B operand: k slots 0..3 = columns 4 lg + r of the chunk's first 16, 4..7 = of its second 16
      if (sh == 0) {
        hf[0] = make_uint4(hm[0].x, hm[0].y, pe.x, pe.y);
        hf[1] = make_uint4(hm[1].x, hm[1].y, pe.z, pe.w);
      } else {
        hf[0] = make_uint4(pe.x, pe.y, hm[0].x, hm[0].y);
        hf[1] = make_uint4(pe.z, pe.w, hm[1].x, hm[1].y);
      }
      const unsigned wq = par * (unsigned)(STAGE / 16) + w2q;
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) {
        const uint4 wf = smem[wq + (unsigned)nt * 64u];  // 16 rows x 64 B further
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc2[nt][mi] = HT<DT>::mfma16(wf, hf[mi], acc2[nt][mi]);
      }
    };

    // first step of the panel: no FF2 yet
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue_next();
    ff1(0);
    ++t;
    FF_TRACE(g, tr, 19);
    // steady state: W1 tile t and W2 slice t - 1 have landed (every DMA a wave issued is older than its wait), the
    // partner's half of chunk t - 1 is in the exchange buffer, every wave is done with the stages about to be refilled
    for (int j = 1; j < NSTEP; ++j, ++t) {
      FF_TRACE(g, tr, 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      FF_TRACE(g, tr, 2);
      asm volatile("s_barrier" ::: "memory");
      FF_TRACE(g, tr, 3);
      issue_next();
      FF_TRACE(g, tr, 4);
      if (!FF_ABLATE(g, 2)) ff2((t + 1u) & 1u, hm_prev);
      __builtin_amdgcn_sched_barrier(0);  // FF1's fragment reads must not be hoisted into FF2 (register pressure)
      FF_TRACE(g, tr, 5);
      if (!FF_ABLATE(g, 2)) ff1(j);
      FF_TRACE(g, tr, 6);
    }
    // drain: FF2 of the panel's last chunk (its W2 slice was issued in the last step)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FF_TRACE(g, tr, 20);
    if constexpr (TAIL) issue_next();  // second projection tile (the first one was issued in the last FF step and has landed)
    ff2((t + 1u) & 1u, hm_prev);
    const int row0 = (int)pr * 32 + li;
    if constexpr (!TAIL) {
      // ---- epilogue: half output, 16-byte stores through the lane exchange of gemm_conv.hip's paired epilogue ----
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) {
        const f32x4 va = acc2[nt][0], vb = acc2[nt][1];
        const auto sx = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[0], va[1]), pack2<DT>(vb[0], vb[1]), false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[2], va[3]), pack2<DT>(vb[2], vb[3]), false, false);
        const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
        // even 16-lane rows: row tile 0, columns 4 lg .. 4 lg + 7; odd rows: row tile 1, columns 4 (lg - 1) ..
        const unsigned row = (unsigned)(row0 + (lg & 1) * 16);
        const unsigned c8 = 160u * sh + 16u * (unsigned)nt + 4u * (unsigned)(lg & ~1);
        __builtin_amdgcn_raw_buffer_store_b128(o, rO, (row * (unsigned)g.ldo + c8) * 2u, 0, MIMO_ST_AUX);
      }
    } else {
      // ---- the block's output projection: out32 = x + z @ Wp^T + bp, z = the FF result just accumulated (fp32, this wave:
      // columns [160 sh, 160 sh + 160)).  z becomes the MFMA operand straight from the accumulators (k order inside a
      // 32-block = the accumulator layout; Wp is packed with the same permutation); the partner's 5 k-steps come through
      // the exchange buffers, one k-step per round. ----
      const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
          (void*)const_cast<float*>(g.x + M0 * g.ldx), 0, (int)(((rows_valid - 1) * g.ldx + C) * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rO32 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out32 + M0 * g.ldo32), 0, (int)(((rows_valid - 1) * g.ldo32 + C) * 4), 0x00020000);
      // (the partner may still be reading this wave's half of the last chunk: no exchange buffer is rewritten before everyone
      // is through its drain)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      exchange_operand([&](int kk, int mi) -> uint4 {  // k-step kk (0..4) of this wave's own columns
        const f32x4 a0 = acc2[2 * kk][mi], a1 = acc2[2 * kk + 1][mi];
        return make_uint4(pack2<DT>(a0[0], a0[1]), pack2<DT>(a0[2], a0[3]), pack2<DT>(a1[0], a1[1]), pack2<DT>(a1[2], a1[3]));
      });
      FF_TRACE(g, tr, 21);
      // accumulators re-initialised with the block input + bias
      const unsigned x_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.ldx + 160 * sh + 4 * lg) * 4));
      const unsigned x_mi = (unsigned)(16 * g.ldx * 4);
#pragma unroll
      for (int nt = 0; nt < 10; ++nt) {
        const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(9 * C / 4 + 4 * nt)]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc2[nt][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, x_off + nt * 64, mi * x_mi, MIMO_LD_AUX)) + bv;
      }
      // position t (tile 0) landed with the drain's wait; every further tile: wait, barrier, issue the next, multiply
      proj(ICf<0>{});
      ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<1>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<2>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<3>{}); ++t;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue_next(); proj(ICf<4>{}); ++t;
      FF_TRACE(g, tr, 22);
      static_assert(NTAIL == 5, "projection tiles are spelled out");
      // (one lane-dependent offset; the row tile travels in the scalar offset, the column tile in the immediate)
      const unsigned o_off = pinned(((unsigned)row0 * (unsigned)g.ldo32 + 160u * sh + 4u * (unsigned)lg) * 4u);
      const unsigned o_mi = 16u * (unsigned)g.ldo32 * 4u;
#pragma unroll
      for (int nt = 0; nt < 10; ++nt)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc2[nt][mi]), rO32,
                                                 o_off + 64u * (unsigned)nt, (unsigned)mi * o_mi, MIMO_ST_AUX);
      // ---- optional: GroupNorm column statistics of out32, per 32-row slab = exactly this wave's rows: (mean, sum of squared
      // deviations from that mean) per column, an exact two-pass computation on the values still in registers (sum over the two
      // row tiles, then over the 16 rows of a tile by DPP, fixed order), in the layout of mimo_gemm_ext's colstats.  The norm
      // that consumes out32 (the motion module's / the next ResBlock's GroupNorm) then makes no statistics pass over HBM. ----
      if (g.colstats) {
        auto dpp = [](float v, auto ctrl_c) {
          return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl_c)::value, 0xf, 0xf, true));
        };
        auto row16_sum = [&](float v) {
          v += dpp(v, ICf<0xB1>{});   // quad_perm [1,0,3,2]
          v += dpp(v, ICf<0x4E>{});   // quad_perm [2,3,0,1]
          v += dpp(v, ICf<0x141>{});  // row_half_mirror
          v += dpp(v, ICf<0x140>{});  // row_mirror
          return v;
        };
        // (one lane-dependent offset made inside the panel loop from values that are live anyway; the panel's slab base and
        // the number of valid slabs travel in the buffer descriptor: a slab beyond M is dropped by the range check)
        const __amdgpu_buffer_rsrc_t rCS = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(g.colstats + (M0 >> 5) * 2 * C), 0, (int)(((rows_valid + 31) >> 5) * 2 * C * 4), 0x00020000);
        const unsigned cs_off = pinned(li == 0 ? (unsigned)((pr * 2 * C + 160 * sh + 4 * lg) * 4) : 0x80000000u);
#pragma unroll
        for (int nt = 0; nt < 10; ++nt) {
          f32x4 s = acc2[nt][0] + acc2[nt][1];
#pragma unroll
          for (int r = 0; r < 4; ++r) s[r] = row16_sum(s[r]);
          const f32x4 mean = s * (1.0f / 32.0f);
          const f32x4 d0 = acc2[nt][0] - mean, d1 = acc2[nt][1] - mean;
          f32x4 q = d0 * d0 + d1 * d1;
#pragma unroll
          for (int r = 0; r < 4; ++r) q[r] = row16_sum(q[r]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mean), rCS, cs_off + 64u * (unsigned)nt, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q), rCS, cs_off + 64u * (unsigned)nt, (unsigned)(C * 4), 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

template <int DT>
static void ff_launch_dt(const FFArgs& g, int mode, unsigned grid, hipStream_t st) {
  if (mode == 4) hipLaunchKernelGGL((ff_fused_kernel<DT, 4>), dim3(grid), dim3(512), 0, st, g);
  else if (mode == 3) hipLaunchKernelGGL((ff_fused_kernel<DT, 3>), dim3(grid), dim3(512), 0, st, g);
  else if (mode == 2) hipLaunchKernelGGL((ff_fused_kernel<DT, 2>), dim3(grid), dim3(512), 0, st, g);
  else if (mode == 1) hipLaunchKernelGGL((ff_fused_kernel<DT, 1>), dim3(grid), dim3(512), 0, st, g);
  else hipLaunchKernelGGL((ff_fused_kernel<DT, 0>), dim3(grid), dim3(512), 0, st, g);
}

#ifdef MIMO_TUNE
extern "C" unsigned long long* mimo_tune_trace_buf();
#endif

// ff_tail4.hip: the block tail (MODE 2) on four waves with 512 registers each
void mimo_ff4_launch(int dtype, const void* args, int mode, unsigned grid, void* stream);
#ifndef MIMO_FF_TAIL4_DEFAULT
#define MIMO_FF_TAIL4_DEFAULT 1
#endif

static int ff_launch(int dtype, FFArgs& g, int mode, void* stream) {
#ifdef MIMO_TUNE
  g.dbg = tune_env("MIMO_FF_TRACE", 0) ? mimo_tune_trace_buf() : nullptr;
  g.ablate = tune_env("MIMO_FF_ABLATE", 0);
#endif
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t npanels = (g.M + BM - 1) / BM;
  const unsigned grid = (unsigned)(npanels < cus ? npanels : cus);
  hipStream_t st = (hipStream_t)stream;
  if (dtype != MIMO_F16 && dtype != MIMO_BF16) return MIMO_EDTYPE;
  // (ff4_kernel also has a feed-forward-only MODE 0 — tools/ff4_variants.py times it — but only its whole-tail form is faster
  // than this file's kernel: 0.626 against 0.674 ms at M = 196 608; MODE 0 0.526 against 0.515)
#ifdef MIMO_FF_TAIL4_MODE0   // the variant builds of tools/ff4_variants.py time ff4_kernel's MODE 0 through mimo_ff_fused as well
  const bool ff4_mode0 = mode == 0;
#else
  const bool ff4_mode0 = false;
#endif
  if ((mode == 2 || ff4_mode0) && tune_env("MIMO_FF_TAIL4", MIMO_FF_TAIL4_DEFAULT)) {
    mimo_ff4_launch(dtype, &g, mode, grid, stream);
    MIMO_LAUNCH_CHECK();
    return MIMO_OK;
  }
  if (dtype == MIMO_F16) ff_launch_dt<MIMO_F16>(g, mode, grid, st);
  else if (dtype == MIMO_BF16) ff_launch_dt<MIMO_BF16>(g, mode, grid, st);
  else return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_ff_fused(int dtype, const void* A, int64_t lda, const void* W1, const float* b1, const void* W2,
                             const float* b2, const float* residual, int64_t ldr, void* out, int64_t ldo, int64_t M,
                             int C_, void* stream) {
  if (!A || !W1 || !W2 || !residual || !out || M <= 0) return MIMO_EINVAL;
  if (C_ != C) return MIMO_EINVAL;  // built for the 320-wide level (K = 320 operand in registers)
  if ((lda & 7) || (ldr & 3) || (ldo & 7) || !aligned16(A) || !aligned16(W1) || !aligned16(W2) || !aligned16(residual) || !aligned16(out))
    return MIMO_EINVAL;
  if (((M - 1) * lda + C) * 2 >= 0x80000000LL || ((M - 1) * ldr + C) * 4 >= 0x100000000LL) return MIMO_EINVAL;
  FFArgs g{};
  g.A = (const uint16_t*)A; g.W1 = (const uint16_t*)W1; g.W2 = (const uint16_t*)W2; g.b1 = b1; g.b2 = b2; g.res = residual;
  g.out = (uint16_t*)out; g.lda = lda; g.ldr = ldr; g.ldo = ldo; g.M = M;
  return ff_launch(dtype, g, 0, stream);
}

extern "C" int mimo_ff_proj_fused(int dtype, const void* A, int64_t lda, const void* W1, const float* b1, const void* W2,
                                  const float* b2, const float* residual, int64_t ldr, const void* Wp, const float* bp,
                                  const float* x, int64_t ldx, float* out, int64_t ldo, int64_t M, int C_, float* colstats,
                                  void* stream) {
  if (!A || !W1 || !W2 || !residual || !Wp || !x || !out || M <= 0) return MIMO_EINVAL;
  if (colstats && ((M & 31) || !aligned16(colstats))) return MIMO_EINVAL;
  if (C_ != C) return MIMO_EINVAL;
  if ((lda & 7) || (ldr & 3) || (ldx & 3) || (ldo & 3) || !aligned16(A) || !aligned16(W1) || !aligned16(W2) || !aligned16(residual) ||
      !aligned16(Wp) || !aligned16(x) || !aligned16(out))
    return MIMO_EINVAL;
  if (((M - 1) * lda + C) * 2 >= 0x80000000LL || ((M - 1) * ldr + C) * 4 >= 0x100000000LL || ((M - 1) * ldx + C) * 4 >= 0x100000000LL ||
      ((M - 1) * ldo + C) * 4 >= 0x100000000LL)
    return MIMO_EINVAL;
  FFArgs g{};
  g.A = (const uint16_t*)A; g.W1 = (const uint16_t*)W1; g.W2 = (const uint16_t*)W2; g.b1 = b1; g.b2 = b2; g.res = residual;
  g.lda = lda; g.ldr = ldr; g.M = M;
  g.Wp = (const uint16_t*)Wp; g.bp = bp; g.x = x; g.out32 = out; g.ldx = ldx; g.ldo32 = ldo; g.colstats = colstats;
  return ff_launch(dtype, g, 1, stream);
}

extern "C" int mimo_block_tail_fused(int dtype, const void* O, int64_t ldo_in, const void* Wstream, const float* bo,
                                     const float* img_bias, int64_t ldib, int64_t rows_per_img, const float* residual,
                                     int64_t ldr, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* b1,
                                     const void* W2, const float* b2, const float* bp, const float* x, int64_t ldx,
                                     float* out, int64_t ldo, int64_t M, int C_, float* colstats, void* stream) {
  if (!O || !Wstream || !residual || !ln_gamma || !ln_beta || !W2 || !x || !out || M <= 0) return MIMO_EINVAL;
  if (colstats && ((M & 31) || !aligned16(colstats))) return MIMO_EINVAL;
  if (C_ != C) return MIMO_EINVAL;
  if (img_bias && (rows_per_img < BM || (ldib & 3) || !aligned16(img_bias))) return MIMO_EINVAL;  // (<= two images per panel)
  if ((ldo_in & 7) || (ldr & 3) || (ldx & 3) || (ldo & 3) || !aligned16(O) || !aligned16(Wstream) || !aligned16(W2) ||
      !aligned16(residual) || !aligned16(x) || !aligned16(out))
    return MIMO_EINVAL;
  if (((M - 1) * ldo_in + C) * 2 >= 0x80000000LL || ((M - 1) * ldr + C) * 4 >= 0x100000000LL || ((M - 1) * ldx + C) * 4 >= 0x100000000LL ||
      ((M - 1) * ldo + C) * 4 >= 0x100000000LL)
    return MIMO_EINVAL;
  if (img_bias && (((M + rows_per_img - 1) / rows_per_img - 1) * ldib + C) * 4 >= 0x80000000LL) return MIMO_EINVAL;
  FFArgs g{};
  g.A = (const uint16_t*)O; g.W1 = (const uint16_t*)Wstream; g.W2 = (const uint16_t*)W2; g.b1 = b1; g.b2 = b2; g.res = residual;
  g.lda = ldo_in; g.ldr = ldr; g.M = M;
  g.bp = bp; g.x = x; g.out32 = out; g.ldx = ldx; g.ldo32 = ldo; g.colstats = colstats;
  g.bo = bo; g.img_bias = img_bias; g.ldib = ldib; g.rows_per_img = img_bias ? rows_per_img : 1;
  g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln_eps = ln_eps;
  return ff_launch(dtype, g, 2, stream);
}

extern "C" int mimo_block_head_fused(int dtype, const void* A, int64_t lda, const float* x32, int64_t ldx, const float* gn_ab,
                                     int64_t rows_per_img, const void* Wstream, const float* bi, const float* residual,
                                     int64_t ldr, const float* ln_gamma, const float* ln_beta, float ln_eps, const float* ln_pe,
                                     int64_t ln_rows_per_frame, int ln_pe_frames, float* y_out, int64_t ldy, void* qkv,
                                     int64_t ldq, int64_t M, int C_, void* stream) {
  if ((!A) == (!x32) || !Wstream || !ln_gamma || !ln_beta || !y_out || !qkv || M <= 0) return MIMO_EINVAL;
  if (C_ != C) return MIMO_EINVAL;
  if (x32 && (!gn_ab || rows_per_img < BM || (ldx & 3) || !aligned16(x32) || !aligned16(gn_ab))) return MIMO_EINVAL;   // (<= two images per panel)
  if (A && ((lda & 7) || !aligned16(A))) return MIMO_EINVAL;
  if (residual && ((ldr & 3) || !aligned16(residual))) return MIMO_EINVAL;
  if (ln_pe && (ln_rows_per_frame < BM || ln_pe_frames <= 0 || !aligned16(ln_pe))) return MIMO_EINVAL;   // (<= two frames per panel)
  if ((ldy & 3) || (ldq & 7) || !aligned16(Wstream) || !aligned16(y_out) || !aligned16(qkv)) return MIMO_EINVAL;
  if ((A && ((M - 1) * lda + C) * 2 >= 0x80000000LL) || (x32 && ((M - 1) * ldx + C) * 4 >= 0x100000000LL) ||
      (residual && ((M - 1) * ldr + C) * 4 >= 0x100000000LL) || ((M - 1) * ldy + C) * 4 >= 0x100000000LL ||
      ((M - 1) * ldq + 3 * C) * 2 >= 0x100000000LL)
    return MIMO_EINVAL;
  FFArgs g{};
  g.A = (const uint16_t*)A; g.lda = lda; g.x = x32; g.ldx = ldx; g.gn_ab = gn_ab; g.rows_per_img = x32 ? rows_per_img : 1;
  g.W1 = (const uint16_t*)Wstream; g.bo = bi; g.res = residual; g.ldr = ldr; g.M = M;
  g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln_eps = ln_eps;
  g.ln_pe = ln_pe; g.ln_rows_per_frame = ln_pe ? ln_rows_per_frame : 1; g.ln_pe_frames = ln_pe ? ln_pe_frames : 1;
  g.out32 = y_out; g.ldo32 = ldy; g.out = (uint16_t*)qkv; g.ldo = ldq;
  return ff_launch(dtype, g, x32 ? 4 : 3, stream);
}
