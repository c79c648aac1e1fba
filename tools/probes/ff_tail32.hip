// ff_tail32.hip — ff_tail4.hip's fused tail of a C = 320 transformer block (four waves, one per SIMD, 512 registers each, wave w
// owns rows [32 w, 32 w + 32) of the 128-row panel and all 320 columns) on v_mfma_f32_32x32x16 instead of 16x16x32.
//
// AN EXPERIMENT, NOT PRODUCT CODE (tools/ff4_variants.py builds it against csrc/ff_fused.hip; the library does not contain it): built, correct (1.7e-5 rel-L2 against ff4_kernel on the level-0
// shape, 256 VGPR + 192 AGPR, no scratch) and measured 10 % SLOWER — 0.708 against 0.637 ms per launch, feed-forward-only entry
// 0.581 against 0.537 (tools/ff4_variants.py `mfma32`, profiles/r6_ff_tail_mfma32_variants.txt).  The idea (round-5 / 6 verdict item
// 3): with one wave per SIMD a 16-cycle 16x16x32 MFMA hides none of the wave's other instructions, a 32-cycle 32x32x16 MFMA hides
// about four, and a skeleton of the position does get faster in shader cycles (tools/probes/ff_position_probe.hip,
// profiles/r6_ff_position_probe.txt: 3 758 -> 3 179 cycles at 12 VALU per segment).  In the kernel every ablation costs MORE than
// its ff4_kernel counterpart, the MFMAs alone + 10 % (`abl32_mfma_only` 0.413 against 0.363 ms): on real data the chip clocks to
// its power budget, and the 32x32x16 form sustains 1.59 PFLOP/s against 1.73-1.82 of 16x16x32 (profiles/r5_mfma_ceiling.txt) —
// the instruction slots it hides are paid for with clock.  Kept as the record of the experiment and for the variant builds.
//
// Same entry point, same packed weights (mimo_amd.packing), same stream protocol and counted waits as ff_tail4.hip — only the
// mapping inside the wave changes:
//
//   lane l = (i = l & 31, h = l >> 5) owns token row 32 w + i.  An accumulator tile c (0..9) = output columns 32 c .. 32 c + 31 is
//   sixteen registers: register 4 b + r = column 32 c + 8 b + 4 h + r.  The half operand of a k-step of 16 is one uint4 per lane:
//   k = 16 ks + 8 h .. + 7 of row i (20 k-steps = 80 registers, as before).
//   A weight fragment = 32 rows x 16 k: lane (i, h) reads row i, 16-byte chunk 2 (ks & 3) + h of K-block ks >> 2.
//   The K permutations the packed weights already carry (pack_ff2_kperm for W2, W1 and Wp in the tail) fit this layout as they are:
//   K position 16 ks + 8 h + e of the consumer = column 32 (ks >> 1) + 16 (e >> 2) + 8 (ks & 1) + 4 h + (e & 3) of the producer, i.e.
//   registers 4 (ks & 1) + (e & 3) and 8 + 4 (ks & 1) + (e & 3) of tile ks >> 1 — no exchange between lanes anywhere.
//   A W1 tile of 64 rows = [16 value | 16 gate] rows of pair 0, then of pair 1: ONE fragment per pair, value and gate of hidden
//   column 16 p + 8 b0 + 4 h + r in registers 4 b0 + r and 8 + 4 b0 + r of the pair's accumulator; a GEGLU unit = (pair, b0).
//   The LDS images are written by the DMA lanes with the swizzles that make THESE fragment reads conflict-free on ds_read_b128's
//   lane groups (W1: chunk ^ ((row >> 1) & 7); W2: chunk ^ ((row >> 2) & 3)): only the lane -> global address map differs.
//   GroupNorm column statistics of the 32-row slab: two columns share a register pair through v_permlane16_swap, then the
//   16-lane row sums of ff_tail4.hip.
#include "ff_fused.hip.h"

#ifndef FF32_FENCE
#define FF32_FENCE 1
#endif
#ifndef FF32_ABLATE   // timing experiments of the feed-forward positions (results are wrong): 1 no DMA issue, 2 no GEGLU VALU,
#define FF32_ABLATE 0 //   4 no MFMAs, 8 no W fragment reads from LDS, 16 no per-position wait + barrier
#endif
#define FF32_SEG_FENCE() do { if (FF32_FENCE) __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {

constexpr int KS16 = C / 16;                // 20 k-steps of the 320-wide operand
constexpr int LDS32_BYTES = XCH_OFF + 256;  // two weight stages + the bias image

template <int A, int B, class F>
__device__ __forceinline__ void static_for32(F&& f) {
  if constexpr (A < B) {
    f(ICf<A>{});
    static_for32<A + 1, B>(f);
  }
}

// block b (registers 4 b .. 4 b + 3) of a 32 x 32 accumulator tile
template <int B_>
__device__ __forceinline__ f32x4 blk(const f32x16& v) {
  return (f32x4){v[4 * B_], v[4 * B_ + 1], v[4 * B_ + 2], v[4 * B_ + 3]};
}
template <int B_>
__device__ __forceinline__ void blk_add(f32x16& v, const f32x4& a) {
  v[4 * B_] += a[0]; v[4 * B_ + 1] += a[1]; v[4 * B_ + 2] += a[2]; v[4 * B_ + 3] += a[3];
}
template <int B_>
__device__ __forceinline__ void blk_set(f32x16& v, const f32x4& a) {
  v[4 * B_] = a[0]; v[4 * B_ + 1] = a[1]; v[4 * B_ + 2] = a[2]; v[4 * B_ + 3] = a[3];
}

template <int DT, int MODE>
__global__ __launch_bounds__(256, 1) void ff32_kernel(const FFArgs g) {
  static_assert(MODE == 0 || MODE == 2, "feed-forward only | whole block tail");
  constexpr bool TAIL = MODE == 2;
  constexpr int NPRE = TAIL ? NTAIL : 0;
  constexpr int NPOS = NPRE + NSTEP + (TAIL ? NTAIL : 0);
  __shared__ __attribute__((aligned(16))) uint4 smem[LDS32_BYTES / 16];  // ONE LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned pr = __builtin_amdgcn_readfirstlane((unsigned)tid >> 6);   // row group: rows [32 pr, 32 pr + 32) of the panel
  const int ri = lane & 31, rh = lane >> 5;
  const unsigned npanels = (unsigned)((g.M + BM - 1) / BM);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  float* const bias_lds = reinterpret_cast<float*>(reinterpret_cast<char*>(&smem[0]) + BIAS_OFF);
  for (int n = tid; n < 8 * C; n += 256) bias_lds[n] = g.b1 ? g.b1[n] : 0.f;
  for (int n = tid; n < C; n += 256) {
    bias_lds[8 * C + n] = g.b2 ? g.b2[n] : 0.f;
    bias_lds[9 * C + n] = (TAIL && g.bp) ? g.bp[n] : 0.f;
    bias_lds[10 * C + n] = (TAIL && g.bo) ? g.bo[n] : 0.f;
    bias_lds[11 * C + n] = TAIL ? g.ln_gamma[n] : 0.f;
    bias_lds[12 * C + n] = TAIL ? g.ln_beta[n] : 0.f;
  }

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const i32x4 rW1 = make_rsrc(g.W1, (unsigned)(TAIL ? NPOS * 64 : 8 * C) * (unsigned)ROWB1);
  const i32x4 rW2 = make_rsrc(g.W2, (unsigned)C * (unsigned)(HID * 2));
  constexpr unsigned OOBA = 0x80000000u;

  // ---- W stream (ff_tail4.hip's: pieces of 1 KB, 0..39 the W1 tile, 40..59 the W2 slice; wave w moves W1 pieces w + 4 i (10) and
  // W2 pieces w + 4 i (5) of every position).  A W1 piece = 8 tile rows (row group w + 4 (i & 1)) x one 128-byte K-block: its lane
  // (row lane >> 3, slot lane & 7) fetches chunk slot ^ ((row >> 1) & 7), (row >> 1) & 7 = (4 (w & 1) + (lane >> 4)) & 7.  A W2 piece
  // = 16 rows x 64 bytes: lane (row lane >> 2, slot lane & 3) fetches chunk slot ^ ((row >> 2) & 3) = slot ^ ((lane >> 4) & 3). ----
  const unsigned w1_lane = ((unsigned)lane >> 3) * (unsigned)ROWB1 +
                           ((((unsigned)lane & 7u) ^ ((4u * (pr & 1u) + ((unsigned)lane >> 4)) & 7u)) << 4);
  const unsigned w2_lane = ((unsigned)lane >> 2) * (unsigned)(HID * 2) + ((((unsigned)lane & 3u) ^ (((unsigned)lane >> 4) & 3u)) << 4);
  auto dma = [&](const i32x4& r_, unsigned voff, unsigned sbase, auto sconst_c, unsigned dbase, auto dconst_c) {
    const i32x4 r = {__builtin_amdgcn_readfirstlane(r_.x), __builtin_amdgcn_readfirstlane(r_.y),
                     __builtin_amdgcn_readfirstlane(r_.z), __builtin_amdgcn_readfirstlane(r_.w)};
    unsigned soff;
    asm volatile("s_add_u32 m0, %4, %5\n\ts_add_u32 %0, %3, %6\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                 : "=&s"(soff)
                 : "v"(voff), "s"(r), "s"(__builtin_amdgcn_readfirstlane(sbase)), "s"(__builtin_amdgcn_readfirstlane(dbase)),
                   "i"(decltype(dconst_c)::value), "i"(decltype(sconst_c)::value)
                 : "memory", "m0", "scc");
  };
  const unsigned my_panels = blockIdx.x < npanels ? (npanels - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
  const unsigned total = my_panels * (unsigned)NPOS;
  unsigned ld_t = 0, ld_j = 0;
  unsigned is_v1 = OOBA, is_s1 = 0, is_d1 = 0, is_s2 = 0, is_d2 = 0;
  bool is_live2 = false;
  auto issue_begin = [&]() {
    const bool live1 = ld_t < total;
    const unsigned j2 = ld_j == 0u ? (unsigned)NPOS - 1u : ld_j - 1u;
    is_live2 = ld_t >= 1u && ld_t <= total && j2 >= (unsigned)NPRE && j2 < (unsigned)(NPRE + NSTEP);
    is_v1 = live1 ? w1_lane : OOBA;
    is_s1 = __builtin_amdgcn_readfirstlane(ld_j * (unsigned)W1_TILE + pr * (8u * ROWB1));
    is_d1 = __builtin_amdgcn_readfirstlane(smem_base + (ld_t & 1u) * (unsigned)STAGE + pr * 1024u);
    is_s2 = __builtin_amdgcn_readfirstlane((j2 - (unsigned)NPRE) * 64u + pr * (16u * HID * 2u));
    is_d2 = __builtin_amdgcn_readfirstlane(smem_base + ((ld_t + 1u) & 1u) * (unsigned)STAGE + (unsigned)W1_TILE + pr * 1024u);
  };
  auto issue_w1 = [&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    dma(rW1, is_v1, is_s1, ICf<(i & 1) * 4 * 8 * ROWB1 + (i >> 1) * 128>{}, is_d1, ICf<i * 4096>{});
  };
  auto issue_w2 = [&](auto i_c) {   // (caller knows the slice is live)
    constexpr int i = decltype(i_c)::value;
    dma(rW2, w2_lane, is_s2, ICf<i * 4 * 16 * HID * 2>{}, is_d2, ICf<i * 4096>{});
  };
  auto issue_end = [&]() {
    ld_t = __builtin_amdgcn_readfirstlane(ld_t + 1u);
    ld_j = __builtin_amdgcn_readfirstlane(ld_j + 1u == (unsigned)NPOS ? 0u : ld_j + 1u);
  };
  auto issue_all = [&]() {   // one burst (positions outside the feed-forward loop)
    issue_begin();
    static_for32<0, 10>([&](auto i_c) { issue_w1(i_c); });
    if (is_live2) static_for32<0, 5>([&](auto i_c) { issue_w2(i_c); });
    issue_end();
  };

  // fragment indices (uint4 units).  W1-region tile: row 32 f + ri, K-block ks >> 2, chunk 2 (ks & 3) + rh, stored at
  // chunk ^ ((row >> 1) & 7): with (ri >> 1) & 7 = 2 u + v the slot of m = ks & 3 is 2 (m ^ u) + (rh ^ v).
  const unsigned fu = ((unsigned)ri >> 2) & 3u, fv = ((unsigned)ri >> 1) & 1u;
  unsigned fq[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) fq[m] = (unsigned)ri * 8u + 2u * ((unsigned)m ^ fu) + ((unsigned)rh ^ fv);
  // W2 slice: row 32 c + ri, chunk 2 s + rh of four, stored at chunk ^ ((row >> 2) & 3)
  unsigned w2q[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) w2q[s] = (unsigned)(W1_TILE / 16) + (unsigned)ri * 4u + ((2u * (unsigned)s + (unsigned)rh) ^ fu);
  constexpr unsigned BIAS_Q = BIAS_OFF / 16;

  issue_all();      // W1-region tile 0
  for (int n = tid; n < 2 * (W2_TILE / 16); n += 256)   // the W2 slices' LDS: finite before the first position multiplies zeros with it
    smem[(n >= W2_TILE / 16 ? STAGE / 16 - W2_TILE / 16 : 0) + W1_TILE / 16 + n] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();  // bias image complete

  // ---- a wave's 32 x 320 slice of the half operand: row 32 pr + ri, k = 16 ks + 8 rh .. + 7 ----
  uint4 fa[KS16];
  const unsigned a_off0 = (unsigned)(((int64_t)(pr * 32 + ri) * g.lda + rh * 8) * 2);
  auto desc_a = [&](int64_t m0) {   // (a panel beyond M: empty descriptor, zeros)
    const int64_t rv = g.M - m0 < (int64_t)BM ? g.M - m0 : (int64_t)BM;
    return __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<uint16_t*>(g.A + (rv > 0 ? m0 : 0) * g.lda), 0,
                                             rv > 0 ? (int)(((rv - 1) * g.lda + C) * 2) : 0, 0x00020000);
  };
  auto desc_r = [&](int64_t m0) {
    const int64_t rv = g.M - m0 < (int64_t)BM ? g.M - m0 : (int64_t)BM;
    return __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.res + (rv > 0 ? m0 : 0) * g.ldr), 0,
                                             rv > 0 ? (int)(((rv - 1) * g.ldr + C) * 4) : 0, 0x00020000);
  };
  // five of the operand's twenty 16-byte loads (part k = 0..3)
  auto load_a_part = [&](uint4 (&dst)[KS16], const __amdgpu_buffer_rsrc_t& rA, auto k_c) {
    const unsigned a_off = pinned(a_off0);
    static_for32<0, 5>([&](auto i_c) {
      constexpr int ks = 5 * decltype(k_c)::value + decltype(i_c)::value;
      dst[ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_off + ks * 32, 0, MIMO_LD_AUX));
    });
  };
  auto load_a = [&](int64_t m0) {
    const __amdgpu_buffer_rsrc_t rA = desc_a(m0);
    static_for32<0, 4>([&](auto k_c) { load_a_part(fa, rA, k_c); });
  };
  // column-tile groups: projection tile q completes tiles q and 5 + q; part k (0..3) of a group = blocks 2 (k & 1), 2 (k & 1) + 1 of
  // tile (k < 2 ? q : 5 + q): two 16-byte loads per lane
  auto load_part = [&](f32x4 (&dst)[10][4], const __amdgpu_buffer_rsrc_t& rs, unsigned off, auto gq_c, auto k_c) {
    constexpr int c = (decltype(k_c)::value < 2 ? 0 : 5) + decltype(gq_c)::value, b0 = 2 * (decltype(k_c)::value & 1);
#pragma unroll
    for (int b = b0; b < b0 + 2; ++b)
      dst[c][b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (32 * c + 8 * b) * 4, 0, MIMO_LD_AUX));
  };
  [[maybe_unused]] uint4 fa_next[KS16];
  [[maybe_unused]] f32x4 opr[10][4];
  [[maybe_unused]] const unsigned r_off0 = (unsigned)(((int64_t)(pr * 32 + ri) * g.ldr + 4 * rh) * 4);
  if constexpr (TAIL) {
    load_a((int64_t)blockIdx.x * BM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (once: the counted waits below assume tile 0 and the first operand have landed)
    const __amdgpu_buffer_rsrc_t rR0 = desc_r((int64_t)blockIdx.x * BM);
    static_for32<0, 4>([&](auto k_c) { load_part(opr, rR0, pinned(r_off0), ICf<0>{}, k_c); });
  }

  unsigned t = 0;
  [[maybe_unused]] unsigned tr = 0;
  for (unsigned panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const int64_t M0 = (int64_t)panel * BM;
    const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
    FF_TRACE(g, tr, 10);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(g.res + M0 * g.ldr), 0, (int)(((rows_valid - 1) * g.ldr + C) * 4), 0x00020000);
    if constexpr (!TAIL) load_a(M0);
    // ---- accumulators: ten tiles of 32 columns, register 4 b + r = column 32 c + 8 b + 4 rh + r of row ri ----
    f32x16 acc2[10];
    const unsigned r_off = pinned((unsigned)(((int64_t)(pr * 32 + ri) * g.ldr + 4 * rh) * 4));
    const unsigned bcol = pinned(BIAS_Q + (unsigned)rh);
    // bias row `row` of the bias image, columns of tile c block b
    auto bias4 = [&](int row, int c, int b) { return __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(row * C / 4 + 8 * c + 2 * b)]); };
    if constexpr (!TAIL) {
      static_for32<0, 10>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        static_for32<0, 4>([&](auto b_c) {
          constexpr int b = decltype(b_c)::value;
          blk_set<b>(acc2[c], __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, r_off + (32 * c + 8 * b) * 4, 0, MIMO_LD_AUX)) +
                                  bias4(8, c, b));
        });
      });
    }
    // one 64-row tile of a [C, C] weight (rows in the tile order of pack_proj_tail: rows 0..31 = output columns 32 q .. + 31, rows
    // 32..63 = 160 + 32 q .. + 31) in the W1 region of stage t & 1, times the operand in fa: ten segments of two k-steps (four MFMAs
    // of 32 cycles).  ISSUE: this wave's pieces of the next position go out in the first five segments (two each), then the hook's
    // loads (segments 5..8), then the tail hook (segment 9): the counted waits rely on this order.
    auto no_hook = [](auto) {};
    auto no_tail = []() {};
    auto proj = [&](auto q_c, auto issue_c, auto&& hook, auto&& tail_hook, auto init_c) {
      constexpr int q = decltype(q_c)::value;
      constexpr bool ISSUE = decltype(issue_c)::value != 0;
      constexpr int INIT = decltype(init_c)::value;   // > 0: the tile's accumulators START from row INIT of the bias image
      static_assert(INIT > 0, "a projection tile's accumulators are born inside the tile");
      static_for32<0, 4>([&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        blk_set<b>(acc2[q], bias4(INIT, q, b));
        blk_set<b>(acc2[5 + q], bias4(INIT, 5 + q, b));
      });
      const unsigned sq = (t & 1u) * (unsigned)(STAGE / 16);
      if constexpr (ISSUE) issue_begin();
      uint4 wf[2][2][2];   // [buffer][k-step of the segment][fragment]
      auto load4 = [&](auto sg_c, uint4 (&dst)[2][2]) {
        constexpr int sg = decltype(sg_c)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const unsigned qq = sq + fq[2 * (sg & 1) + kk] + (unsigned)((sg >> 1) * 512);
          dst[kk][0] = smem[qq];
          dst[kk][1] = smem[qq + 256u];
        }
      };
      load4(ICf<0>{}, wf[0]);
      static_for32<0, 10>([&](auto sg_c) {
        constexpr int sg = decltype(sg_c)::value;
        if constexpr (sg + 1 < 10) load4(ICf<sg + 1>{}, wf[(sg + 1) & 1]);
        if constexpr (ISSUE && sg < 5) { issue_w1(ICf<2 * sg>{}); issue_w1(ICf<2 * sg + 1>{}); }
        if constexpr (sg >= 5 && sg < 9) hook(ICf<sg - 5>{});
        if constexpr (sg == 9) tail_hook();   // (behind the hook's loads: the counted waits rely on the order)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          acc2[q] = HT<DT>::mfma32(wf[sg & 1][kk][0], fa[2 * sg + kk], acc2[q]);
          acc2[5 + q] = HT<DT>::mfma32(wf[sg & 1][kk][1], fa[2 * sg + kk], acc2[5 + q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (ISSUE) issue_end();
    };
    // the accumulators as the next MFMA operand (the consuming weight carries the K permutation, see the file header)
    auto acc_to_operand = [&](auto&& f) {   // f(c_c, b_c) -> f32x4
      static_for32<0, KS16>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value, c = ks >> 1, par = ks & 1;
        const f32x4 a0 = f(ICf<c>{}, ICf<par>{}), a1 = f(ICf<c>{}, ICf<2 + par>{});
        fa[ks] = make_uint4(pack2<DT>(a0[0], a0[1]), pack2<DT>(a0[2], a0[3]), pack2<DT>(a1[0], a1[1]), pack2<DT>(a1[2], a1[3]));
      });
    };

    if constexpr (TAIL) {
      // the per-image vector (the collapsed cross-attention of the spatial blocks): rows_per_img >= 128, a panel holds rows of at
      // most two images; a lane owns ONE row
      auto add_img_bias = [&]() {
        if (g.img_bias) {
          const int64_t nimg = (g.M + g.rows_per_img - 1) / g.rows_per_img;
          const __amdgpu_buffer_rsrc_t rIB = __builtin_amdgcn_make_buffer_rsrc(
              (void*)const_cast<float*>(g.img_bias), 0, (int)(((nimg - 1) * g.ldib + C) * 4), 0x00020000);
          const int64_t img0 = M0 / g.rows_per_img;
          const int next0 = (int)((img0 + 1) * g.rows_per_img - M0);
          const unsigned ib_off = (unsigned)(((img0 + ((int)(pr * 32) + ri >= next0 ? 1 : 0)) * g.ldib + 4 * rh) * 4);   // (rows past M: zeros)
          // ten loads in flight, then their additions
          static_for32<0, 4>([&](auto qt_c) {
            constexpr int c0 = decltype(qt_c)::value < 2 ? 0 : 5, bb = 2 * (decltype(qt_c)::value & 1);
            f32x4 v[10];
            static_for32<0, 10>([&](auto i_c) {
              constexpr int i = decltype(i_c)::value, c = c0 + i / 2, b = bb + (i & 1);
              v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIB, ib_off + (32 * c + 8 * b) * 4, 0, 0));
            });
            __builtin_amdgcn_sched_barrier(0);
            static_for32<0, 10>([&](auto i_c) {
              constexpr int i = decltype(i_c)::value, c = c0 + i / 2, b = bb + (i & 1);
              blk_add<b>(acc2[c], v[i]);
            });
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      };
      // y = residual + o @ Wo^T + bo: five tiles of Wo (positions 0..4 of the panel).  The residual's column-tile groups 1..4 are
      // fetched under tiles 0..3 (group 0 went out at the panel start); group q is added once tile q is through: at most three
      // groups (96 registers) in flight.  Counted waits as in ff_tail4.hip (the per-wave operation counts are the same).
      auto add_res = [&](auto gq_c) {
        constexpr int q = decltype(gq_c)::value;
        static_for32<0, 4>([&](auto b_c) {
          constexpr int b = decltype(b_c)::value;
          blk_add<b>(acc2[q], opr[q][b]);
          blk_add<b>(acc2[5 + q], opr[5 + q][b]);
        });
      };
      asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 11);
      proj(ICf<0>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, ICf<1>{}, k_c); }, no_tail, ICf<10>{}); ++t;
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 12);
      proj(ICf<1>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, ICf<2>{}, k_c); }, [&]() { add_res(ICf<0>{}); }, ICf<10>{}); ++t;
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 13);
      proj(ICf<2>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, ICf<3>{}, k_c); }, [&]() { add_res(ICf<1>{}); }, ICf<10>{}); ++t;
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 14);
      proj(ICf<3>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, ICf<4>{}, k_c); }, [&]() { add_res(ICf<2>{}); }, ICf<10>{}); ++t;
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 15);
      proj(ICf<4>{}, ICf<1>{}, no_hook, [&]() { add_res(ICf<3>{}); }, ICf<10>{}); ++t;
      add_res(ICf<4>{});
      add_img_bias();
      // LayerNorm over the row's 320 columns: per half of 160 columns (tiles 0..4 | 5..9) a local sum and a local centred sum of
      // squares (this lane's 80 values + the partner lane's, l ^ 32), combined by the pairwise update
      float mean, rstd;
      {
        float hs[2], hq[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          float sum = 0.f;
#pragma unroll
          for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              sum += (acc2[5 * hh + c][4 * b] + acc2[5 * hh + c][4 * b + 1]) + (acc2[5 * hh + c][4 * b + 2] + acc2[5 * hh + c][4 * b + 3]);
          sum += __shfl_xor(sum, 32, 64);
          const float ml = sum * (1.f / 160.f);
          float qq = 0.f;
#pragma unroll
          for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = acc2[5 * hh + c][r] - ml;
              qq = fmaf(d, d, qq);
            }
          qq += __shfl_xor(qq, 32, 64);
          hs[hh] = sum; hq[hh] = qq;
        }
        const float dm = (hs[0] - hs[1]) * (1.f / 160.f);
        mean = (hs[0] + hs[1]) * (1.f / 320.f);
        rstd = rsqrtf(((hq[0] + hq[1]) + 80.f * dm * dm) * (1.f / 320.f) + g.ln_eps);
      }
      acc_to_operand([&](auto c_c, auto b_c) -> f32x4 {
        constexpr int c = decltype(c_c)::value, b = decltype(b_c)::value;
        return (blk<b>(acc2[c]) - mean) * rstd * bias4(11, c, b) + bias4(12, c, b);
      });
      // the feed-forward accumulates on y + b2
      static_for32<0, 10>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        static_for32<0, 4>([&](auto b_c) { blk_add<decltype(b_c)::value>(acc2[c], bias4(8, c, decltype(b_c)::value)); });
      });
    }

    // ---- feed-forward, 40 positions.  A position = 30 SEGMENTS of two 32-cycle MFMAs: 0..9 FF1 of the W1 tile's value/gate pair 0
    // (two k-steps per segment), 10..19 FF1 of pair 1, 20..29 FF2 of the PREVIOUS chunk (column tile s, both k-steps).  A segment is
    // fenced: the two W fragments of segment s + 2 are fetched from LDS, at most one DMA piece of the next position goes out, the
    // MFMAs are issued, and one stage of the GEGLU units in flight runs in their shadow.  A GEGLU unit = (pair, b0): the four
    // value / gate registers 4 b0 + r | 8 + 4 b0 + r of the pair's accumulator = hidden columns 16 p + 8 b0 + 4 rh + r. ----
    f32x16 acc1[2];
    struct GU { float x[4], u[4], v[4], p[4]; u32x2 out; };
    auto gu_stage = [&](GU& s, auto st_c, const f32x4& val, const f32x4& gate, const f32x4& bval, const f32x4& bgate) {
      constexpr int st = decltype(st_c)::value;   // gelu_erf_f (common.hip.h) spread over ten stages
      if constexpr ((FF32_ABLATE & 2) != 0) {
        if constexpr (st == 9) { s.out.x = __float_as_uint(val[0]); s.out.y = __float_as_uint(gate[0]); }
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (st == 0) {
          s.x[r] = gate[r] + bgate[r];
          s.v[r] = val[r] + bval[r];
          s.u[r] = __builtin_amdgcn_fmed3f(__builtin_fabsf(s.x[r]), 0.f, 6.0f);
        } else if constexpr (st == 1) {
          s.p[r] = fmaf(GELU_P[7], s.u[r], GELU_P[6]);
        } else if constexpr (st <= 6) {
          s.p[r] = fmaf(s.p[r], s.u[r], GELU_P[7 - st]);
        } else if constexpr (st == 7) {
          s.p[r] = fmaf(s.p[r], s.u[r], GELU_P[0]);
          s.x[r] = fmaxf(s.x[r], 0.f);
        } else if constexpr (st == 8) {
          s.p[r] = __builtin_amdgcn_exp2f(s.p[r]);
        } else {
          s.p[r] = s.v[r] * fmaf(-s.u[r], s.p[r], s.x[r]);
        }
      }
      if constexpr (st == 9) {
        s.out.x = pack2<DT>(s.p[0], s.p[1]);
        s.out.y = pack2<DT>(s.p[2], s.p[3]);
      }
    };
    auto gu_all = [&](GU& s, const f32x4& val, const f32x4& gate, const f32x4& bval, const f32x4& bgate) {
      static_for32<0, 10>([&](auto st_c) { gu_stage(s, st_c, val, gate, bval, bgate); });
    };
    // b1 of chunk j, unit (p, b0): value bias | gate bias of hidden columns 32 j + 16 p + 8 b0 + 4 rh + r
    auto bias_val = [&](unsigned j, int p, int b0) { return __builtin_bit_cast(f32x4, smem[bcol + 16u * j + (unsigned)(8 * p + 2 * b0)]); };
    auto bias_gate = [&](unsigned j, int p, int b0) { return __builtin_bit_cast(f32x4, smem[bcol + 16u * j + (unsigned)(8 * p + 2 * b0 + 4)]); };
    auto frag_load = [&](auto seg_c, unsigned sq, unsigned wq, uint4 (&dst)[2]) {
      constexpr int seg = decltype(seg_c)::value;
      if constexpr ((FF32_ABLATE & 8) != 0) {
        dst[0] = make_uint4(sq, wq, 1u, 2u); dst[1] = make_uint4(wq, sq, 3u, 4u);
      } else if constexpr (seg < 20) {
        constexpr int p = seg / 10, kk2 = seg % 10;   // k-steps 2 kk2, 2 kk2 + 1: K-block kk2 >> 1, m = 2 (kk2 & 1), + 1
        const unsigned qq = sq + (unsigned)((kk2 >> 1) * 512 + p * 256);
        dst[0] = smem[qq + fq[2 * (kk2 & 1)]];
        dst[1] = smem[qq + fq[2 * (kk2 & 1) + 1]];
      } else {
        constexpr int c = seg - 20;
        dst[0] = smem[wq + w2q[0] + (unsigned)(c * 128)];
        dst[1] = smem[wq + w2q[1] + (unsigned)(c * 128)];
      }
    };
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto seg_mma = [&](auto seg_c, const uint4 (&src)[2], const uint4 (&hf)[2]) {
      constexpr int seg = decltype(seg_c)::value;
      if constexpr ((FF32_ABLATE & 4) != 0) {
        if constexpr (seg % 10 == 0 && seg < 20) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[seg / 10][r] = __builtin_bit_cast(float, r & 1 ? src[0].x : src[1].y);
        }
      } else if constexpr (seg < 20) {
        constexpr int p = seg / 10, kk2 = seg % 10;
        acc1[p] = HT<DT>::mfma32(src[0], fa[2 * kk2], kk2 == 0 ? zero16 : acc1[p]);
        acc1[p] = HT<DT>::mfma32(src[1], fa[2 * kk2 + 1], acc1[p]);
      } else {
        constexpr int c = seg - 20;
        acc2[c] = HT<DT>::mfma32(src[0], hf[0], acc2[c]);
        acc2[c] = HT<DT>::mfma32(src[1], hf[1], acc2[c]);
      }
    };

    u32x2 hA[2], hB[2];          // the previous chunk, packed: pair 0 / pair 1, per b0 (= k-step of FF2)
    f32x4 cv, cg, cbv, cbg;      // unit (pair 1, b0 = 1) of the previous chunk and its b1 values: it runs under the next position's
                                 // first segments
    // The first position has no previous chunk: it runs the same code on an all-zero one.
    hA[0] = hA[1] = hB[0] = hB[1] = (u32x2){0u, 0u};
    cv = cg = cbv = cbg = zero4;
    FF_TRACE(g, tr, 18);
#pragma unroll 1
    for (unsigned j = 0; j < (unsigned)NSTEP; ++j, ++t) {
      if constexpr ((FF32_ABLATE & 16) == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      FF_TRACE(g, tr, 1);
      const unsigned sq = (t & 1u) * (unsigned)(STAGE / 16);
      const unsigned wq = ((t + 1u) & 1u) * (unsigned)(STAGE / 16);
      issue_begin();
      uint4 fr[3][2];
      uint4 hf[2] = {};
      frag_load(ICf<0>{}, sq, wq, fr[0]);
      frag_load(ICf<1>{}, sq, wq, fr[1]);
      GU u1, u2, u3, u4;
      f32x4 bv00 = zero4, bg00 = zero4, bv01 = zero4, bg01 = zero4, bv10 = zero4, bg10 = zero4, bv11 = zero4, bg11 = zero4;
      __builtin_amdgcn_sched_barrier(0);
      static_for32<0, 30>([&](auto seg_c) {
        constexpr int seg = decltype(seg_c)::value;
        if constexpr (seg + 2 < 30) frag_load(ICf<seg + 2>{}, sq, wq, fr[(seg + 2) % 3]);
        if constexpr (seg == 9) { bv00 = bias_val(j, 0, 0); bg00 = bias_gate(j, 0, 0); bv01 = bias_val(j, 0, 1); bg01 = bias_gate(j, 0, 1); }
        if constexpr (seg == 19) { bv10 = bias_val(j, 1, 0); bg10 = bias_gate(j, 1, 0); bv11 = bias_val(j, 1, 1); bg11 = bias_gate(j, 1, 1); }
        if constexpr ((FF32_ABLATE & 1) == 0) {
          if constexpr (seg < 10) issue_w1(seg_c);
          else if constexpr (seg < 20 && (seg & 1) != 0) issue_w2(ICf<(seg - 10) / 2>{});
        }
        if constexpr (seg == 20) {
          hf[0] = make_uint4(hA[0].x, hA[0].y, hB[0].x, hB[0].y);
          hf[1] = make_uint4(hA[1].x, hA[1].y, hB[1].x, hB[1].y);
        }
        seg_mma(seg_c, fr[seg % 3], hf);
        if constexpr (seg < 10) {
          gu_stage(u1, seg_c, cv, cg, cbv, cbg);                       // chunk j - 1, unit (1, 1)
          if constexpr (seg == 9) hB[1] = u1.out;
        } else {
          if constexpr (seg < 20) gu_stage(u2, ICf<seg - 10>{}, blk<0>(acc1[0]), blk<2>(acc1[0]), bv00, bg00);   // chunk j, unit (0, 0)
          if constexpr ((seg & 1) == 0) gu_stage(u3, ICf<(seg - 10) / 2>{}, blk<1>(acc1[0]), blk<3>(acc1[0]), bv01, bg01);   // unit (0, 1)
          if constexpr (seg >= 20) gu_stage(u4, ICf<seg - 20>{}, blk<0>(acc1[1]), blk<2>(acc1[1]), bv10, bg10);   // unit (1, 0)
        }
        if constexpr (seg == 19) issue_end();
        FF32_SEG_FENCE();
      });
      hA[0] = u2.out; hA[1] = u3.out; hB[0] = u4.out;
      cv = blk<1>(acc1[1]); cg = blk<3>(acc1[1]); cbv = bv11; cbg = bg11;
    }
    // drain: FF2 of the panel's last chunk (its W2 slice was issued in the last step)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FF_TRACE(g, tr, 20);
    if constexpr (TAIL) issue_all();  // second projection tile (the first one was issued in the last FF step and has landed)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(TAIL ? g.x + M0 * g.ldx : nullptr), 0, TAIL ? (int)(((rows_valid - 1) * g.ldx + C) * 4) : 0, 0x00020000);
    [[maybe_unused]] const unsigned x_off = pinned((unsigned)(((int64_t)(pr * 32 + ri) * g.ldx + 4 * rh) * 4));
    if constexpr (TAIL) {   // x, column-tile group 0 (behind the DMA pieces just issued; groups 1..4 under the projection tiles)
      static_for32<0, 4>([&](auto k_c) { load_part(opr, rX, x_off, ICf<0>{}, k_c); });
    }
    {
      GU u1;
      gu_all(u1, cv, cg, cbv, cbg);
      hB[1] = u1.out;
      const uint4 hf[2] = {make_uint4(hA[0].x, hA[0].y, hB[0].x, hB[0].y), make_uint4(hA[1].x, hA[1].y, hB[1].x, hB[1].y)};
      const unsigned wq = ((t + 1u) & 1u) * (unsigned)(STAGE / 16);
      uint4 fr[2][2];
      frag_load(ICf<20>{}, 0u, wq, fr[0]);
      static_for32<20, 30>([&](auto seg_c) {
        constexpr int seg = decltype(seg_c)::value;
        if constexpr (seg + 1 < 30) frag_load(ICf<seg + 1>{}, 0u, wq, fr[(seg + 1) & 1]);
        seg_mma(seg_c, fr[seg & 1], hf);
      });
    }
    const int row0 = (int)pr * 32 + ri;
    if constexpr (!TAIL) {
      // ---- half output: 8-byte stores (columns 32 c + 8 b + 4 rh .. + 3 of row ri) ----
      const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out + M0 * g.ldo), 0, (int)(((rows_valid - 1) * g.ldo + C) * 2), 0x00020000);
      const unsigned o_off = ((unsigned)row0 * (unsigned)g.ldo + 4u * (unsigned)rh) * 2u;
      static_for32<0, 10>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        static_for32<0, 4>([&](auto b_c) {
          constexpr int b = decltype(b_c)::value;
          const f32x4 v = blk<b>(acc2[c]);
          const u32x2 o = {pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])};
          __builtin_amdgcn_raw_buffer_store_b64(o, rO, o_off + (unsigned)((32 * c + 8 * b) * 2), 0, MIMO_ST_AUX);
        });
      });
    } else {
      // ---- the block's output projection: out32 = x + z @ Wp^T + bp, z = the feed-forward result in the accumulators ----
      const __amdgpu_buffer_rsrc_t rO32 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out32 + M0 * g.ldo32), 0, (int)(((rows_valid - 1) * g.ldo32 + C) * 4), 0x00020000);
      acc_to_operand([&](auto c_c, auto b_c) -> f32x4 { return blk<decltype(b_c)::value>(acc2[decltype(c_c)::value]); });
      __builtin_amdgcn_sched_barrier(0);   // (the new accumulators must not be created while the old ones are still being packed)
      FF_TRACE(g, tr, 23);
      const unsigned o_off = pinned(((unsigned)row0 * (unsigned)g.ldo32 + 4u * (unsigned)rh) * 4u);
      const __amdgpu_buffer_rsrc_t rCS = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.colstats ? g.colstats + (M0 >> 5) * 2 * C : nullptr), 0, g.colstats ? (int)(((rows_valid + 31) >> 5) * 2 * C * 4) : 0, 0x00020000);
      // the lane that stores a column quadruple's statistics: lane 0 of each 16-lane row; rows 0 / 2 hold blocks b' (columns
      // 8 b' + 4 rh ..), rows 1 / 3 blocks b' + 2
      const unsigned cs_off = pinned((lane & 15) == 0 ? (unsigned)((pr * 2 * C + 16 * ((lane >> 4) & 1) + 4 * rh) * 4) : 0x80000000u);
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
      // swap(a, b): lanes of 16-lane rows 0 / 2 keep a and receive b of the row below... -> {[a.r0, b.r0, a.r2, b.r2], [a.r1, b.r1, a.r3, b.r3]}
      auto swap16 = [](float a, float b, float& lo, float& hi) {
        const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
        lo = __builtin_bit_cast(float, s[0]);
        hi = __builtin_bit_cast(float, s[1]);
      };
      // one finished column tile leaves: the output row (four 16-byte stores per lane) and the GroupNorm column statistics of the
      // 32-row slab = this wave's rows (layout of ff_tail4.hip / mimo_gemm_ext's colstats: mean, then the sum of squared deviations
      // from it; fixed order).  Without a statistics buffer the descriptor is empty: the stores are dropped.
      auto store_tile = [&](auto c_c, const f32x4 (&v)[4]) {
        constexpr int c = decltype(c_c)::value;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[b]), rO32, o_off + (unsigned)((32 * c + 8 * b) * 4), 0, MIMO_ST_AUX);
#pragma unroll
        for (int bp_ = 0; bp_ < 2; ++bp_) {
          f32x4 mean_c, qv, ma, mb;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float lo, hi;
            swap16(v[bp_][r], v[bp_ + 2][r], lo, hi);
            mean_c[r] = row16_sum(lo + hi) * (1.0f / 32.0f);
            float ma_, mb_;
            swap16(mean_c[r], mean_c[r], ma_, mb_);   // -> the mean of block b' | of block b' + 2 on both rows of the half
            ma[r] = ma_; mb[r] = mb_;
          }
          const f32x4 d0 = v[bp_] - ma, d1 = v[bp_ + 2] - mb;
          const f32x4 q0 = d0 * d0, q1 = d1 * d1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float lo, hi;
            swap16(q0[r], q1[r], lo, hi);
            qv[r] = row16_sum(lo + hi);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mean_c), rCS, cs_off + (unsigned)((32 * c + 8 * bp_) * 4), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), rCS, cs_off + (unsigned)((32 * c + 8 * bp_) * 4), (unsigned)(C * 4), 0);
        }
      };
      // a finished column-tile group: + x, out (8 output stores + 8 statistics stores)
      auto finish = [&](auto gq_c) {
        constexpr int q = decltype(gq_c)::value;
        static_for32<0, 2>([&](auto f_c) {
          constexpr int c = 5 * decltype(f_c)::value + q;
          const f32x4 v[4] = {blk<0>(acc2[c]) + opr[c][0], blk<1>(acc2[c]) + opr[c][1], blk<2>(acc2[c]) + opr[c][2], blk<3>(acc2[c]) + opr[c][3]};
          store_tile(ICf<c>{}, v);
        });
      };
      // x group q is fetched under tile q - 1 (group 0 at the drain) and used when tile q + 1 is nearly through: at most three
      // groups in flight.  Counted waits, behind the pieces of the tile about to be read: tile 1 — x groups 0, 1: 16 loads;
      // tiles 2, 3, 4 — 8 loads + the 16 stores of a finished group.
      proj(ICf<0>{}, ICf<0>{}, [&](auto k_c) { load_part(opr, rX, x_off, ICf<1>{}, k_c); }, no_tail, ICf<9>{}); ++t;
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 24);
      proj(ICf<1>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, ICf<2>{}, k_c); }, [&]() { finish(ICf<0>{}); }, ICf<9>{}); ++t;
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 25);
      proj(ICf<2>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, ICf<3>{}, k_c); }, [&]() { finish(ICf<1>{}); }, ICf<9>{}); ++t;
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 26);
      proj(ICf<3>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, ICf<4>{}, k_c); }, [&]() { finish(ICf<2>{}); }, ICf<9>{}); ++t;
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 27);
      // under the last tile: the NEXT panel's half operand (20 loads, behind the pieces of its tile 0, in front of this panel's
      // last 32 stores: nothing at the next panel's start has to wait for a store)
      const int64_t M0n = M0 + (int64_t)gridDim.x * BM;
      const __amdgpu_buffer_rsrc_t rAn = desc_a(M0n);
      proj(ICf<4>{}, ICf<1>{}, [&](auto k_c) { load_a_part(fa_next, rAn, k_c); }, [&]() { finish(ICf<3>{}); }, ICf<9>{}); ++t;
      FF_TRACE(g, tr, 28);
      finish(ICf<4>{});
      {   // the next panel's residual, column-tile group 0
        const __amdgpu_buffer_rsrc_t rRn = desc_r(M0n);
        static_for32<0, 4>([&](auto k_c) { load_part(opr, rRn, pinned(r_off0), ICf<0>{}, k_c); });
      }
      // an empty asm that READS the next operand: the compiler places its own counted wait for the 20 loads in front of it (it
      // knows the 40 operations it issued behind them) and from here on treats them as landed
      asm volatile("" :: "v"(__builtin_bit_cast(u32x4, fa_next[0])), "v"(__builtin_bit_cast(u32x4, fa_next[1])), "v"(__builtin_bit_cast(u32x4, fa_next[2])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[3])), "v"(__builtin_bit_cast(u32x4, fa_next[4])), "v"(__builtin_bit_cast(u32x4, fa_next[5])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[6])), "v"(__builtin_bit_cast(u32x4, fa_next[7])), "v"(__builtin_bit_cast(u32x4, fa_next[8])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[9])), "v"(__builtin_bit_cast(u32x4, fa_next[10])), "v"(__builtin_bit_cast(u32x4, fa_next[11])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[12])), "v"(__builtin_bit_cast(u32x4, fa_next[13])), "v"(__builtin_bit_cast(u32x4, fa_next[14])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[15])), "v"(__builtin_bit_cast(u32x4, fa_next[16])), "v"(__builtin_bit_cast(u32x4, fa_next[17])),
                   "v"(__builtin_bit_cast(u32x4, fa_next[18])), "v"(__builtin_bit_cast(u32x4, fa_next[19])) : "memory");
#pragma unroll
      for (int ks = 0; ks < KS16; ++ks) fa[ks] = fa_next[ks];
    }
  }
  FF_TRACE(g, tr, 30);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
  FF_TRACE(g, tr, 31);
}

template <int DT>
void ff32_launch_dt(const FFArgs& g, int mode, unsigned grid, hipStream_t st) {
  if (mode == 2) hipLaunchKernelGGL((ff32_kernel<DT, 2>), dim3(grid), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((ff32_kernel<DT, 0>), dim3(grid), dim3(256), 0, st, g);
}

}  // namespace

// called by ff_launch (ff_fused.hip) for MODE 0 / 2
#ifdef FF32_AS_FF4   // tools/ff4_variants.py: this file in the place of ff_tail4.hip
void mimo_ff4_launch(int dtype, const void* args, int mode, unsigned grid, void* stream) {
#else
void mimo_ff32_launch(int dtype, const void* args, int mode, unsigned grid, void* stream) {
#endif
  const FFArgs& g = *static_cast<const FFArgs*>(args);
  if (dtype == MIMO_F16) ff32_launch_dt<MIMO_F16>(g, mode, grid, (hipStream_t)stream);
  else ff32_launch_dt<MIMO_BF16>(g, mode, grid, (hipStream_t)stream);
}
