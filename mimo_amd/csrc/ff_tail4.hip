// ff_tail4.hip — the fused tail of a C = 320 transformer block (mimo_block_tail_fused: ff_fused.hip's MODE 2; MODE 0 exists for the
// variant builds) on FOUR waves, one per SIMD, each with the whole 512-entry register file (gfx950).  Same entry point, same packed
// weights, same arithmetic as ff_fused_kernel, a different machine mapping:
//
//   ff_fused_kernel (8 waves x 256 registers): the two waves of a SIMD share 32 rows and split the 320 columns; every hidden
//   chunk, every LayerNorm statistic and every re-used accumulator tile crosses between them through LDS behind barriers.
//
//   ff4_kernel (this file): wave w owns rows [32 w, 32 w + 32) of the 128-row panel and ALL 320 columns: 160 accumulator
//   registers (the compiler places them in AGPRs) + 80 operand registers + 32 FF1 accumulators.  Nothing is exchanged: the
//   GEGLU chunk, the LayerNorm operand and the projection operand go from accumulator to MFMA operand inside the wave; the only
//   barrier is the one per stream position.
//
// What was measured on it (NOTEBOOK.md round 5 §13, profiles/r5_ff4_*):
//   * a 16-cycle 16x16x32 MFMA hides NONE of its own wave's VALU / LDS / DMA instructions (a feed-forward position costs
//     120 x 16 + 4 x the other instructions), exactly as two waves per SIMD hide none of each other's: the feed-forward is
//     written as fenced SEGMENTS of four MFMAs with one stage of the GEGLU polynomial each (so the order is mine, not the
//     scheduler's), but what counts is the instruction count: a DMA piece is three instructions, W fragments are fetched two
//     segments ahead;
//   * a third of a panel's time used to be HBM bursts: all 256 persistent blocks reach the panel boundary together.  With
//     FF4_TRICKLE (shipped) the fp32 operands are added BEHIND the projections they used to initialise, fetched eight loads per
//     projection tile behind that tile's DMA pieces (at most three column-tile groups in flight), a projection tile's
//     accumulators are born inside the tile, a finished column-tile group is stored under the next tile, and the next panel's
//     half operand is fetched under the last tile.  vmcnt is ONE in-order counter for loads, stores and DMA pieces: every
//     position barrier below waits with the COUNT of what this wave issued behind the pieces it is about to read — keep the
//     issue order (pieces, then the hook's loads, then the tail hook's stores) when touching the projection code.
//   FF4_TRICKLE = 0 keeps ff_fused_kernel's operand order and is bit-identical to it (tools/ff4_variants.py, `exact`).
//
// Stream protocol (positions, 2-deep ring, one barrier per position, W2 slice one position behind its W1 tile) is the one of
// ff_fused.hip; the weight tensors are the ones mimo_amd.packing already makes.
#include "ff_fused.hip.h"

// Compile-time experiment knobs (tools/ff4_variants.py builds one small library per setting; the shipped library uses the defaults)
#ifndef FF4_ABLATE    // timing experiments of the feed-forward positions (results are wrong): 1 no DMA issue, 2 no GEGLU VALU,
#define FF4_ABLATE 0  //   4 no MFMAs, 8 no W fragment reads from LDS, 16 no per-position wait + barrier
#endif
#ifndef FF4_FENCE     // 1: sched_barrier between segments
#define FF4_FENCE 1
#endif
#ifndef FF4_PF        // W fragments are fetched this many segments ahead (1 | 2)
#define FF4_PF 2
#endif
#ifndef FF4_BIAS_INIT // 1: the FF1 accumulators start from b1 instead of zero (the GEGLU's two bias additions disappear; the sum
#define FF4_BIAS_INIT 0 //  is then b + sum(a w) instead of sum(a w) + b: not bit-identical to ff_fused_kernel any more)
#endif
#ifndef FF4_TRICKLE   // 1 (MODE 2): the fp32 operands are added BEHIND the projections they used to initialise (y = (o Wo^T + bo) + res,
#define FF4_TRICKLE 1 //    out = (z Wp^T + bp) + x) and their loads trickle in under the projection tiles, eight per tile, behind that tile's
#endif                //    DMA pieces (counted vmcnt waits); the next panel's half operand is fetched before this panel's stores go out.
                      //    0: the operand order of ff_fused_kernel (bit-identical to it; tools/ff4_variants.py `exact`)
#ifndef FF4_DEPHASE   // > 0: every other group of 8 blocks starts this many thousand cycles late.  All blocks walk their panels in
#define FF4_DEPHASE 0 //    lockstep, so the chip alternates between phases with no HBM traffic at all (the feed-forward positions)
#endif                //    and HBM-bound bursts (operand loads / stores at the panel boundary); two groups half a panel apart halve the bursts
#ifndef FF4_TOUCH     // 1: during the feed-forward positions the waves touch (4-byte LDS-DMA loads into a junk LDS row, one 128-byte line
#define FF4_TOUCH 0   //    per lane) the NEXT panel's operand rows and this panel's x rows: every block reaches its panel boundary at the
#endif                //    same time, and the burst of operand loads there is an HBM-bound stall of the whole chip
#define FF4_LD_BYTES(x) (x)
#define FF4_LD_ROW0(m0) (((FF4_ABLATE) & 32) ? (int64_t)0 : (m0))   // 32: every panel loads the operand rows of panel 0 (cache hits, same statistics)
#define FF4_ST_BYTES(x) (((FF4_ABLATE) & 64) ? 0 : (x))   // 64: no output stores (dropped by the range check)
#define FF4_SEG_FENCE() do { if (FF4_FENCE) __builtin_amdgcn_sched_barrier(0); } while (0)

namespace {

constexpr int LDS4_JUNK = XCH_OFF;    // 256 bytes nobody reads: the destination of the touch loads
constexpr int LDS4_BYTES = XCH_OFF + 256;   // two weight stages + the bias image (no exchange buffers)

// compile-time loop: f(ICf<I>{}) for I in [A, B) — every register-array index below is a constant expression
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) {
    f(ICf<A>{});
    static_for<A + 1, B>(f);
  }
}

template <int DT, int MODE>
__global__ __launch_bounds__(256, 1) void ff4_kernel(const FFArgs g) {
  static_assert(MODE == 0 || MODE == 2, "feed-forward only | whole block tail");
  constexpr bool TAIL = MODE == 2;
  constexpr int NPRE = TAIL ? NTAIL : 0;
  constexpr int NPOS = NPRE + NSTEP + (TAIL ? NTAIL : 0);
  __shared__ __attribute__((aligned(16))) uint4 smem[LDS4_BYTES / 16];  // ONE LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned pr = __builtin_amdgcn_readfirstlane((unsigned)tid >> 6);   // row group: rows [32 pr, 32 pr + 32) of the panel
  const int lg = lane >> 4, li = lane & 15;
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

  // ---- W stream: the LDS images and the piece -> lane mapping of ff_fused.hip (pieces of 1 KB: 0..39 the W1 tile, 40..59 the
  // W2 slice); wave w moves W1 pieces w, w + 4, ... (10) and W2 pieces w, w + 4, ... (5) of every position ----
  const unsigned w1_lane = ((unsigned)lane >> 3) * (unsigned)ROWB1 + ((((unsigned)lane & 7u) ^ (((unsigned)lane >> 3) & 7u)) << 4);
  const unsigned w2_lane = ((unsigned)lane >> 2) * (unsigned)(HID * 2) + ((((unsigned)lane & 3u) ^ (2u * (((unsigned)lane >> 5) & 1u))) << 4);
  // One LDS-DMA piece in THREE instructions: m0 = LDS base + constant; soffset = stream base + constant (the SALU instruction
  // between the m0 write and the DMA is the wait state the hardware asks for); the DMA.  All lane- / wave-dependent parts of the
  // addresses are loop-invariant (voff, the per-wave bases below); the piece index only contributes compile-time constants.
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
  // the position being issued (ld_t; ld_j inside its panel): W1 tile -> W1 stage ld_t & 1, W2 slice of position ld_t - 1 -> W2
  // stage (ld_t - 1) & 1.  begin() fixes the scalars, w1(i) / w2(i) issue one piece each, end() advances.
  // Wave w moves W1 pieces p = w + 4 i (K-block i >> 1, row group w + 4 (i & 1)) and W2 pieces d = w + 4 i:
  //   W1 piece: source offset (w + 4 (i & 1)) * 8 rows + (i >> 1) * 128 B, LDS offset (w + 4 i) KB;  W2 piece: 16 d rows, (w + 4 i) KB
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
    static_for<0, 10>([&](auto i_c) { issue_w1(i_c); });
    if (is_live2) static_for<0, 5>([&](auto i_c) { issue_w2(i_c); });
    issue_end();
  };

  // touch(base, row0, ld_bytes, row_bytes, k): lane i of the k-th touch instruction of a [128 x row_bytes] panel starting at row
  // row0 fetches 4 bytes of line (64 k + i) of the panel (lines counted row by row); rows are clamped to the tensor
  [[maybe_unused]] auto touch = [&](const void* base, int64_t row0, int64_t ld_bytes, unsigned row_bytes, unsigned k) {
    const unsigned lpr = row_bytes >> 7;                       // 128-byte lines per row (5 | 10)
    const unsigned tl = k * 64u + (unsigned)lane;
    const unsigned row = tl / lpr, line = tl - row * lpr;
    int64_t r = row0 + (int64_t)row;
    r = r < g.M ? r : g.M - 1;
    const char* ptr = static_cast<const char*>(base) + r * ld_bytes + (int64_t)line * 128;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                 :: "v"(ptr), "s"(__builtin_amdgcn_readfirstlane(smem_base + (unsigned)LDS4_JUNK)) : "memory", "m0");
  };

  // fragment indices (uint4 units), as ff_fused.hip: W1-region tile row 16 n4 + li, logical chunk 4 ks + lg
  const unsigned bq0 = (unsigned)li * 8u + (unsigned)(lg ^ (li & 7));
  const unsigned bq1 = (unsigned)li * 8u + (unsigned)((4 + lg) ^ (li & 7));
  const unsigned w2q = (unsigned)(W1_TILE / 16) + (unsigned)li * 4u + (unsigned)(lg ^ (2 * ((li >> 3) & 1)));   // W2 row 16 nt + li
  constexpr unsigned BIAS_Q = BIAS_OFF / 16;

  issue_all();      // W1-region tile 0
  for (int n = tid; n < 2 * (W2_TILE / 16); n += 256)   // the W2 slices' LDS: finite before the first position multiplies zeros with it
    smem[(n >= W2_TILE / 16 ? STAGE / 16 - W2_TILE / 16 : 0) + W1_TILE / 16 + n] = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (FF4_DEPHASE > 0) {
    if ((blockIdx.x >> 3) & 1u) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < (unsigned long long)FF4_DEPHASE * 1000ull) __builtin_amdgcn_s_sleep(64);
    }
  }
  __syncthreads();  // bias image complete

  constexpr bool TRICKLE = TAIL && FF4_TRICKLE != 0;
  // ---- a wave's 32 x 320 slice of the half operand in MFMA layout: row 32 pr + 16 mi + li, k = 32 ks + 8 lg .. + 7 ----
  uint4 fa[2][KS];
  const unsigned a_off0 = (unsigned)(((int64_t)(pr * 32 + li) * g.lda + lg * 8) * 2);
  const unsigned a_mi = (unsigned)(16 * g.lda * 2);
  auto desc_a = [&](int64_t m0) {   // (a panel beyond M: empty descriptor, zeros)
    const int64_t rv = g.M - m0 < (int64_t)BM ? g.M - m0 : (int64_t)BM;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<uint16_t*>(g.A + (rv > 0 ? FF4_LD_ROW0(m0) : 0) * g.lda), 0, rv > 0 ? FF4_LD_BYTES((int)(((rv - 1) * g.lda + C) * 2)) : 0, 0x00020000);
  };
  auto desc_r = [&](int64_t m0) {
    const int64_t rv = g.M - m0 < (int64_t)BM ? g.M - m0 : (int64_t)BM;
    return __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(g.res + (rv > 0 ? FF4_LD_ROW0(m0) : 0) * g.ldr), 0, rv > 0 ? FF4_LD_BYTES((int)(((rv - 1) * g.ldr + C) * 4)) : 0, 0x00020000);
  };
  // five of the operand's twenty 16-byte loads (part k = 0..3)
  auto load_a_part = [&](uint4 (&dst)[2][KS], const __amdgpu_buffer_rsrc_t& rA, auto k_c) {
    const unsigned a_off = pinned(a_off0);   // (pinned HERE: the ten `offset + constant` must not become ten hoisted registers)
    static_for<0, 5>([&](auto i_c) {
      constexpr int i = 5 * decltype(k_c)::value + decltype(i_c)::value, mi = i / KS, ks = i % KS;
      dst[mi][ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_off + ks * 64, mi * a_mi, MIMO_LD_AUX));
    });
  };
  auto load_a = [&](int64_t m0) {
    const __amdgpu_buffer_rsrc_t rA = desc_a(m0);
    static_for<0, 4>([&](auto k_c) { load_a_part(fa, rA, k_c); });
  };
  // column-tile groups: projection tile q completes column tiles 2 q, 2 q + 1, 10 + 2 q, 11 + 2 q; part k (0..3) of a group = one of
  // them, both row tiles (two 16-byte loads per lane)
  auto load_part = [&](f32x4 (&dst)[20][2], const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned off_mi, auto gq_c, auto k_c) {
    constexpr int nt = (decltype(k_c)::value < 2 ? 0 : 10) + 2 * decltype(gq_c)::value + (decltype(k_c)::value & 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      dst[nt][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + nt * 64, mi * off_mi, MIMO_LD_AUX));
  };
  // TRICKLE: what a panel needs first — its half operand and column-tile group 0 of the residual — is fetched at the END of the
  // previous panel (the operand under its last projection tile, into fa_next); opr = the fp32 operand that is added behind a
  // projection (the residual, later x), at most three column-tile groups of it in flight
  [[maybe_unused]] uint4 fa_next[2][KS];
  [[maybe_unused]] f32x4 opr[20][2];
  [[maybe_unused]] const unsigned r_off0 = (unsigned)(((int64_t)(pr * 32 + li) * g.ldr + 4 * lg) * 4);
  [[maybe_unused]] const unsigned r_mi0 = (unsigned)(16 * g.ldr * 4);
  if constexpr (TRICKLE) {
    load_a((int64_t)blockIdx.x * BM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (once: the counted waits below assume tile 0 and the first operand have landed)
    const __amdgpu_buffer_rsrc_t rR0 = desc_r((int64_t)blockIdx.x * BM);
    static_for<0, 4>([&](auto k_c) { load_part(opr, rR0, pinned(r_off0), r_mi0, ICf<0>{}, k_c); });
  }

  unsigned t = 0;
  [[maybe_unused]] unsigned tr = 0;   // tune build: thread 0 of block 0 stamps the cycle counter behind every position's barrier
  for (unsigned panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const int64_t M0 = (int64_t)panel * BM;
    const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
    FF_TRACE(g, tr, 10);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(g.res + FF4_LD_ROW0(M0) * g.ldr), 0, FF4_LD_BYTES((int)(((rows_valid - 1) * g.ldr + C) * 4)), 0x00020000);
    if constexpr (!TAIL) load_a(M0);
    // ---- accumulators: lane (li, lg) owns columns 16 nt + 4 lg .. + 3 of rows 16 mi + li for all 20 column tiles ----
    f32x4 acc2[20][2];
    const unsigned r_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.ldr + 4 * lg) * 4));
    const unsigned bcol = pinned(BIAS_Q + (unsigned)lg);
    const unsigned r_mi = (unsigned)(16 * g.ldr * 4);
    if constexpr (TRICKLE) {
      // (residual group 0 is on its way; the accumulators start from bo inside the projection tiles)
    } else {
#pragma unroll
      for (int nt = 0; nt < 20; ++nt) {
        const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)((TAIL ? 10 : 8) * C / 4 + 4 * nt)]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc2[nt][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, r_off + nt * 64, mi * r_mi, MIMO_LD_AUX)) + bv;
      }
    }
    // one 64-row tile of a [C, C] weight (rows in the tile order of pack_proj_tail: n-tiles 0, 1 = output columns 32 q .. + 31,
    // n-tiles 2, 3 = 160 + 32 q .. + 31) in the W1 region of stage t & 1, times the operand in fa.  ISSUE: this wave's pieces of
    // the next position go out between the k-steps.
    auto no_hook = [](auto) {};
    auto no_tail = []() {};
    auto proj = [&](auto q_c, auto issue_c, auto&& hook, auto&& tail_hook, auto init_c) {
      constexpr int q = decltype(q_c)::value;
      constexpr bool ISSUE = decltype(issue_c)::value != 0;
      constexpr int INIT = decltype(init_c)::value;   // > 0: the tile's accumulators START from row INIT of the bias image (they
                                                      // come alive here: nothing else has touched this tile's columns yet)
      f32x4 binit[4];
      if constexpr (INIT > 0) {
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
          binit[n4] = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(INIT * C / 4 + 4 * (10 * (n4 >> 1) + 2 * q + (n4 & 1)))]);
      }
      const unsigned sq = (t & 1u) * (unsigned)(STAGE / 16);
      if constexpr (ISSUE) issue_begin();
      uint4 wf[2][4];
      auto load4 = [&](auto ks_c, uint4 (&dst)[4]) {
        constexpr int ks = decltype(ks_c)::value;
        const unsigned qq = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 512);
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) dst[n4] = smem[qq + (unsigned)n4 * 128u];
      };
      load4(ICf<0>{}, wf[0]);
      // segments of eight MFMAs (one k-step), the next k-step's fragments fetched one segment ahead; the DMA pieces of the next
      // position in the first five segments (two each), then the hook's loads (k-steps 5..8: counted waits rely on this order)
      static_for<0, KS>([&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks + 1 < KS) load4(ICf<ks + 1>{}, wf[(ks + 1) & 1]);
        if constexpr (ISSUE && ks < 5) { issue_w1(ICf<2 * ks>{}); issue_w1(ICf<2 * ks + 1>{}); }
        if constexpr (ks >= 5 && ks < 9) hook(ICf<ks - 5>{});
        if constexpr (ks == 9) tail_hook();   // (behind the hook's loads: the counted waits rely on the order)
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc2[10 * (n4 >> 1) + 2 * q + (n4 & 1)][mi] =
                HT<DT>::mfma16(wf[ks & 1][n4], fa[mi][ks], (INIT > 0 && ks == 0) ? binit[n4] : acc2[10 * (n4 >> 1) + 2 * q + (n4 & 1)][mi]);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (ISSUE) {
        if constexpr (!TAIL) {   // (MODE 2 never has a live W2 slice behind a projection position)
          if (is_live2) static_for<0, 5>([&](auto i_c) { issue_w2(i_c); });
        }
        issue_end();
      }
    };
    // the accumulators as the next MFMA operand (k order inside a 32-block = the accumulator layout, the consuming weight is
    // packed with pack_ff2_kperm): k-step 5 h + kk = column tiles 10 h + 2 kk, 10 h + 2 kk + 1
    auto acc_to_operand = [&](auto&& f) {   // f(nt, mi) -> f32x4
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int n0 = 10 * (ks / 5) + 2 * (ks % 5);
          const f32x4 a0 = f(n0, mi), a1 = f(n0 + 1, mi);
          fa[mi][ks] = make_uint4(pack2<DT>(a0[0], a0[1]), pack2<DT>(a0[2], a0[3]), pack2<DT>(a1[0], a1[1]), pack2<DT>(a1[2], a1[3]));
        }
    };

    if constexpr (TAIL) {
      // the per-image vector (the collapsed cross-attention of the spatial blocks): rows_per_img >= 128, a panel holds rows of at
      // most two images
      auto add_img_bias = [&]() {
        if (g.img_bias) {
          const int64_t nimg = (g.M + g.rows_per_img - 1) / g.rows_per_img;
          const __amdgpu_buffer_rsrc_t rIB = __builtin_amdgcn_make_buffer_rsrc(
              (void*)const_cast<float*>(g.img_bias), 0, (int)(((nimg - 1) * g.ldib + C) * 4), 0x00020000);
          const int64_t img0 = M0 / g.rows_per_img;
          const int next0 = (int)((img0 + 1) * g.rows_per_img - M0);
          const unsigned ib_off = (unsigned)((img0 * g.ldib + 4 * lg) * 4);
          const unsigned step = (unsigned)(g.ldib * 4);   // (rows past M read a vector past the table: zeros)
          // ten loads in flight, then their additions (written as one loop the loads are issued and waited for one by one)
          auto batch = [&](auto mi_c, auto half_c, unsigned o, bool both) {
            constexpr int mi = decltype(mi_c)::value, n0 = 10 * decltype(half_c)::value;
            f32x4 v[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rIB, o + (n0 + i) * 64, 0, 0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 10; ++i) {
              acc2[n0 + i][mi] += v[i];
              if (both) acc2[n0 + i][1] += v[i];
            }
            __builtin_amdgcn_sched_barrier(0);
          };
          if (next0 >= BM) {   // the whole panel lies in one image (the pipeline's case): both row tiles take the same vector
            batch(ICf<0>{}, ICf<0>{}, ib_off, true);
            batch(ICf<0>{}, ICf<1>{}, ib_off, true);
          } else {
            const unsigned o0 = ib_off + ((int)(pr * 32) + li >= next0 ? step : 0u);
            const unsigned o1 = ib_off + ((int)(pr * 32) + 16 + li >= next0 ? step : 0u);
            batch(ICf<0>{}, ICf<0>{}, o0, false);
            batch(ICf<0>{}, ICf<1>{}, o0, false);
            batch(ICf<1>{}, ICf<0>{}, o1, false);
            batch(ICf<1>{}, ICf<1>{}, o1, false);
          }
        }
      };
      if constexpr (!TRICKLE) {
        add_img_bias();
        __builtin_amdgcn_sched_barrier(0);
        load_a(M0);
      }
      // y = residual + o @ Wo^T + bo: five tiles of Wo (positions 0..4 of the panel).  TRICKLE: the residual's column-tile groups
      // 1..4 are fetched under tiles 0..3 (group 0 went out at the panel start); group q is added once tile q is through: at most
      // three groups (96 registers) are in flight.
      // Counted waits: vmcnt is ONE in-order counter.  Behind the DMA pieces of the tile about to be read this wave has issued,
      // at tile 0: the operand's 20 loads, the 32 stores of the previous panel's groups 3 and 4, 8 residual loads (the operand
      // itself was waited for at the end of the previous panel: 40 leaves it and the pieces complete); at tiles 1..4: the 8
      // loads of the hook.
      auto add_res = [&](auto gq_c) {
        static_for<0, 4>([&](auto k_c) {
          constexpr int nt = (decltype(k_c)::value < 2 ? 0 : 10) + 2 * decltype(gq_c)::value + (decltype(k_c)::value & 1);
          acc2[nt][0] += opr[nt][0];
          acc2[nt][1] += opr[nt][1];
        });
      };
      if constexpr (TRICKLE) {
        asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 11);
        proj(ICf<0>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, r_mi, ICf<1>{}, k_c); }, no_tail, ICf<10>{}); ++t;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 12);
        proj(ICf<1>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, r_mi, ICf<2>{}, k_c); }, [&]() { add_res(ICf<0>{}); }, ICf<10>{}); ++t;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 13);
        proj(ICf<2>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, r_mi, ICf<3>{}, k_c); }, [&]() { add_res(ICf<1>{}); }, ICf<10>{}); ++t;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 14);
        proj(ICf<3>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rR, r_off, r_mi, ICf<4>{}, k_c); }, [&]() { add_res(ICf<2>{}); }, ICf<10>{}); ++t;
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 15);
        proj(ICf<4>{}, ICf<1>{}, no_hook, [&]() { add_res(ICf<3>{}); }, ICf<10>{}); ++t;
        add_res(ICf<4>{});
        add_img_bias();
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 11); proj(ICf<0>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 12); proj(ICf<1>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 13); proj(ICf<2>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 14); proj(ICf<3>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 15); proj(ICf<4>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
      }
      // LayerNorm over the row's 320 columns, with the arithmetic of ff_fused_kernel (which holds the row in two halves of 160
      // columns on two waves): per half a local sum and a local centred sum of squares, combined by the pairwise update
      float mean[2], rstd[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        float hs[2], hq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float sum = 0.f;
#pragma unroll
          for (int nt = 0; nt < 10; ++nt)
            sum += (acc2[10 * h + nt][mi][0] + acc2[10 * h + nt][mi][1]) + (acc2[10 * h + nt][mi][2] + acc2[10 * h + nt][mi][3]);
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float ml = sum * (1.f / 160.f);
          float qq = 0.f;
#pragma unroll
          for (int nt = 0; nt < 10; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = acc2[10 * h + nt][mi][r] - ml;
              qq = fmaf(d, d, qq);
            }
          qq += __shfl_xor(qq, 16, 64);
          qq += __shfl_xor(qq, 32, 64);
          hs[h] = sum; hq[h] = qq;
        }
        const float dm = (hs[0] - hs[1]) * (1.f / 160.f);
        mean[mi] = (hs[0] + hs[1]) * (1.f / 320.f);
        rstd[mi] = rsqrtf(((hq[0] + hq[1]) + 80.f * dm * dm) * (1.f / 320.f) + g.ln_eps);
      }
      acc_to_operand([&](int nt, int mi) -> f32x4 {
        const f32x4 gm = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(11 * C / 4 + 4 * nt)]);
        const f32x4 bt = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(12 * C / 4 + 4 * nt)]);
        return (acc2[nt][mi] - mean[mi]) * rstd[mi] * gm + bt;
      });
      // the feed-forward accumulates on y + b2
#pragma unroll
      for (int nt = 0; nt < 20; ++nt) {
        const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(8 * C / 4 + 4 * nt)]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc2[nt][mi] += bv;
      }
    }

    // ---- feed-forward, 40 positions.  A position = 30 SEGMENTS of four MFMAs: 0..9 FF1 of the W1 tile's value/gate pair 0
    // (k-step = segment), 10..19 FF1 of pair 1, 20..29 FF2 of the PREVIOUS chunk (column tiles 2 s, 2 s + 1).  A segment is
    // fenced (sched_barrier): the two W fragments of segment s + 2 are fetched from LDS, at most one DMA piece of the next
    // position goes out, the four MFMAs are issued, and one stage of the GEGLU units that are in flight runs in their shadow.
    // A GEGLU unit = one 16 x 16 value / gate tile pair (4 values per lane), 10 stages of 4-12 VALU instructions. ----
    f32x4 acc1[4][2];
    struct GU { float x[4], u[4], v[4], p[4]; u32x2 out; };
    auto gu_stage = [&](GU& s, auto st_c, const f32x4& val, const f32x4& gate, const f32x4& bval, const f32x4& bgate) {
      // gelu_erf_f (common.hip.h) spread over stages; scalar VALU on purpose (packed fp32 ops are slow beside MFMAs)
      constexpr int st = decltype(st_c)::value;
      if constexpr ((FF4_ABLATE & 2) != 0) {
        if constexpr (st == 9) { s.out.x = __float_as_uint(val[0]); s.out.y = __float_as_uint(gate[0]); }
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (st == 0) {
          s.x[r] = FF4_BIAS_INIT ? gate[r] : gate[r] + bgate[r];
          s.v[r] = FF4_BIAS_INIT ? val[r] : val[r] + bval[r];
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
      static_for<0, 10>([&](auto st_c) { gu_stage(s, st_c, val, gate, bval, bgate); });
    };
    // b1 of chunk j, pair p: value bias | gate bias of hidden columns 32 j + 16 p + 4 lg + r
    auto bias_val = [&](unsigned j, int p) { return __builtin_bit_cast(f32x4, smem[BIAS_Q + 16u * j + (unsigned)(8 * p) + (unsigned)lg]); };
    auto bias_gate = [&](unsigned j, int p) { return __builtin_bit_cast(f32x4, smem[BIAS_Q + 16u * j + (unsigned)(8 * p + 4) + (unsigned)lg]); };
    auto frag_load = [&](auto seg_c, unsigned sq, unsigned wq, uint4 (&dst)[2]) {
      constexpr int seg = decltype(seg_c)::value;
      if constexpr ((FF4_ABLATE & 8) != 0) {
        dst[0] = make_uint4(sq, wq, 1u, 2u); dst[1] = make_uint4(wq, sq, 3u, 4u);
      } else if constexpr (seg < 20) {
        const int p = seg / 10, ks = seg % 10;
        const unsigned qq = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 512);
        dst[0] = smem[qq + (unsigned)(2 * p) * 128u];
        dst[1] = smem[qq + (unsigned)(2 * p + 1) * 128u];
      } else {
        const int nt = 2 * (seg - 20);
        dst[0] = smem[wq + (unsigned)nt * 64u];
        dst[1] = smem[wq + (unsigned)(nt + 1) * 64u];
      }
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto seg_mma = [&](auto seg_c, const uint4 (&src)[2], const uint4 (&hf)[2], const f32x4& iv, const f32x4& ig) {
      constexpr int seg = decltype(seg_c)::value;
      if constexpr ((FF4_ABLATE & 4) != 0) {
        if constexpr (seg % 10 == 0 && seg < 20) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) { acc1[2 * (seg / 10)][mi] = __builtin_bit_cast(f32x4, src[0]); acc1[2 * (seg / 10) + 1][mi] = __builtin_bit_cast(f32x4, src[1]); }
        }
      } else if constexpr (seg < 20) {
        const int p = seg / 10, ks = seg % 10;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          acc1[2 * p][mi] = HT<DT>::mfma16(src[0], fa[mi][ks], ks == 0 ? (FF4_BIAS_INIT ? iv : zero4) : acc1[2 * p][mi]);
          acc1[2 * p + 1][mi] = HT<DT>::mfma16(src[1], fa[mi][ks], ks == 0 ? (FF4_BIAS_INIT ? ig : zero4) : acc1[2 * p + 1][mi]);
        }
      } else {
        const int nt = 2 * (seg - 20);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc2[nt + i][mi] = HT<DT>::mfma16(src[i], hf[mi], acc2[nt + i][mi]);
      }
    };

    u32x2 hA[2], hB[2];          // the previous chunk, packed: pair 0 / pair 1, per row tile
    f32x4 cv, cg, cbv, cbg;      // pair 1, rows 16..31 of the previous chunk and its b1 values: that GEGLU unit runs under the
                                 // next position's first segments
    // The first position has no previous chunk: it runs the same code on an all-zero one (zero hidden values times the finite
    // contents of the W2 stage — zero-filled at kernel start, stale weights later — add exactly nothing to the accumulators).
    hA[0] = hA[1] = hB[0] = hB[1] = (u32x2){0u, 0u};
    cv = cg = cbv = cbg = zero4;
    FF_TRACE(g, tr, 18);
#pragma unroll 1
    for (unsigned j = 0; j < (unsigned)NSTEP; ++j, ++t) {
      if constexpr ((FF4_ABLATE & 16) == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      FF_TRACE(g, tr, 1);
      const unsigned sq = (t & 1u) * (unsigned)(STAGE / 16);
      const unsigned wq = ((t + 1u) & 1u) * (unsigned)(STAGE / 16) + w2q;
      issue_begin();
      uint4 fr[3][2];
      uint4 hf[2] = {};
      frag_load(ICf<0>{}, sq, wq, fr[0]);
      if constexpr (FF4_PF == 2) frag_load(ICf<1>{}, sq, wq, fr[1]);
      GU u1, u2, u3, u4;
      f32x4 bv0 = zero4, bg0 = zero4, bv1 = zero4, bg1 = zero4;
      if constexpr (FF4_BIAS_INIT != 0) { bv0 = bias_val(j, 0); bg0 = bias_gate(j, 0); }
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 30>([&](auto seg_c) {
        constexpr int seg = decltype(seg_c)::value;
        if constexpr (seg + FF4_PF < 30) frag_load(ICf<seg + FF4_PF>{}, sq, wq, fr[(seg + FF4_PF) % 3]);
        if constexpr (FF4_BIAS_INIT != 0) {
          if constexpr (seg == 5) { bv1 = bias_val(j, 1); bg1 = bias_gate(j, 1); }
        } else {
          if constexpr (seg == 9) { bv0 = bias_val(j, 0); bg0 = bias_gate(j, 0); }
          if constexpr (seg == 19) { bv1 = bias_val(j, 1); bg1 = bias_gate(j, 1); }
        }
        if constexpr ((FF4_ABLATE & 1) == 0) {
          if constexpr (seg < 10) issue_w1(seg_c);
          else if constexpr (seg < 20 && (seg & 1) != 0) issue_w2(ICf<(seg - 10) / 2>{});
        }
        if constexpr (FF4_TOUCH != 0 && seg == 24) {
          // wave pr issues touch number 4 (j - 1) + pr: 0..9 the next panel's A rows (640 lines), 10..29 its residual rows
          // (1280 lines), MODE 2: 30..49 this panel's x rows
          const unsigned ti = (j - 1u) * 4u + pr;
          const int64_t Mn = M0 + (int64_t)gridDim.x * BM;
          if (ti < 10u) { if (Mn < g.M) touch(g.A, Mn, g.lda * 2, (unsigned)(C * 2), ti); }
          else if (ti < 30u) { if (Mn < g.M) touch(g.res, Mn, g.ldr * 4, (unsigned)(C * 4), ti - 10u); }
          else if (TAIL && ti < 50u) touch(g.x, M0, g.ldx * 4, (unsigned)(C * 4), ti - 30u);
        }
        if constexpr (seg == 20) {
          hf[0] = make_uint4(hA[0].x, hA[0].y, hB[0].x, hB[0].y);
          hf[1] = make_uint4(hA[1].x, hA[1].y, hB[1].x, hB[1].y);
        }
        seg_mma(seg_c, fr[seg % 3], hf, seg < 10 ? bv0 : bv1, seg < 10 ? bg0 : bg1);
        if constexpr (seg < 10) {
          gu_stage(u1, seg_c, cv, cg, cbv, cbg);                       // chunk j - 1, pair 1, rows 16..31
          if constexpr (seg == 9) hB[1] = u1.out;
        } else {
          if constexpr (seg < 20) gu_stage(u2, ICf<seg - 10>{}, acc1[0][0], acc1[1][0], bv0, bg0);   // chunk j, pair 0, rows 0..15
          if constexpr ((seg & 1) == 0) gu_stage(u3, ICf<(seg - 10) / 2>{}, acc1[0][1], acc1[1][1], bv0, bg0);   // chunk j, pair 0, rows 16..31
          if constexpr (seg >= 20) gu_stage(u4, ICf<seg - 20>{}, acc1[2][0], acc1[3][0], bv1, bg1);   // chunk j, pair 1, rows 0..15
        }
        if constexpr (seg == 19) issue_end();
        FF4_SEG_FENCE();
      });
      hA[0] = u2.out; hA[1] = u3.out; hB[0] = u4.out;
      cv = acc1[2][1]; cg = acc1[3][1]; cbv = bv1; cbg = bg1;
    }
    // drain: FF2 of the panel's last chunk (its W2 slice was issued in the last step)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FF_TRACE(g, tr, 20);
    if constexpr (TAIL) issue_all();  // second projection tile (the first one was issued in the last FF step and has landed)
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<float*>(TAIL ? g.x + FF4_LD_ROW0(M0) * g.ldx : nullptr), 0, TAIL ? FF4_LD_BYTES((int)(((rows_valid - 1) * g.ldx + C) * 4)) : 0, 0x00020000);
    [[maybe_unused]] const unsigned x_off = pinned((unsigned)(((int64_t)(pr * 32 + li) * g.ldx + 4 * lg) * 4));
    [[maybe_unused]] const unsigned x_mi = (unsigned)(16 * g.ldx * 4);
    if constexpr (TRICKLE) {   // x, column-tile group 0 (behind the DMA pieces just issued; groups 1..4 under the projection tiles)
      static_for<0, 4>([&](auto k_c) { load_part(opr, rX, x_off, x_mi, ICf<0>{}, k_c); });
    }
    {
      GU u1;
      gu_all(u1, cv, cg, cbv, cbg);
      hB[1] = u1.out;
      const uint4 hf[2] = {make_uint4(hA[0].x, hA[0].y, hB[0].x, hB[0].y), make_uint4(hA[1].x, hA[1].y, hB[1].x, hB[1].y)};
      const unsigned wq = ((t + 1u) & 1u) * (unsigned)(STAGE / 16) + w2q;
      uint4 fr[2][2];
      frag_load(ICf<20>{}, 0u, wq, fr[0]);
      static_for<20, 30>([&](auto seg_c) {
        constexpr int seg = decltype(seg_c)::value;
        if constexpr (seg + 1 < 30) frag_load(ICf<seg + 1>{}, 0u, wq, fr[(seg + 1) & 1]);
        seg_mma(seg_c, fr[seg & 1], hf, zero4, zero4);
      });
    }
    const int row0 = (int)pr * 32 + li;
    if constexpr (!TAIL) {
      // ---- half output, 16-byte stores through the lane exchange of gemm_conv.hip's paired epilogue ----
      const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out + M0 * g.ldo), 0, FF4_ST_BYTES((int)(((rows_valid - 1) * g.ldo + C) * 2)), 0x00020000);
#pragma unroll
      for (int nt = 0; nt < 20; ++nt) {
        const f32x4 va = acc2[nt][0], vb = acc2[nt][1];
        const auto sx = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[0], va[1]), pack2<DT>(vb[0], vb[1]), false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(pack2<DT>(va[2], va[3]), pack2<DT>(vb[2], vb[3]), false, false);
        const u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
        const unsigned row = (unsigned)(row0 + (lg & 1) * 16);
        const unsigned c8 = 16u * (unsigned)nt + 4u * (unsigned)(lg & ~1);
        __builtin_amdgcn_raw_buffer_store_b128(o, rO, (row * (unsigned)g.ldo + c8) * 2u, 0, MIMO_ST_AUX);
      }
    } else {
      // ---- the block's output projection: out32 = x + z @ Wp^T + bp, z = the feed-forward result in the accumulators ----
      const __amdgpu_buffer_rsrc_t rO32 = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.out32 + M0 * g.ldo32), 0, FF4_ST_BYTES((int)(((rows_valid - 1) * g.ldo32 + C) * 4)), 0x00020000);
      acc_to_operand([&](int nt, int mi) -> f32x4 { return acc2[nt][mi]; });
      __builtin_amdgcn_sched_barrier(0);   // (the new accumulators must not be created while the old ones are still being packed)
      if constexpr (!TRICKLE) {   // (TRICKLE: the accumulators start from bp inside the projection tiles)
#pragma unroll
        for (int nt = 0; nt < 20; ++nt) {
          const f32x4 bv = __builtin_bit_cast(f32x4, smem[bcol + (unsigned)(9 * C / 4 + 4 * nt)]);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc2[nt][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, x_off + nt * 64, mi * x_mi, MIMO_LD_AUX)) + bv;
        }
      }
      // position t (tile 0) landed with the drain's wait; every further tile: wait, barrier, multiply (+ issue the next)
      FF_TRACE(g, tr, 23);
      const unsigned o_off = pinned(((unsigned)row0 * (unsigned)g.ldo32 + 4u * (unsigned)lg) * 4u);
      const unsigned o_mi = 16u * (unsigned)g.ldo32 * 4u;
      const __amdgpu_buffer_rsrc_t rCS = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(g.colstats ? g.colstats + (M0 >> 5) * 2 * C : nullptr), 0, g.colstats ? FF4_ST_BYTES((int)(((rows_valid + 31) >> 5) * 2 * C * 4)) : 0, 0x00020000);
      const unsigned cs_off = pinned(li == 0 ? (unsigned)((pr * 2 * C + 4 * lg) * 4) : 0x80000000u);
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
      // one finished column tile leaves: the output rows (two 16-byte stores per lane) and, optionally, the GroupNorm column
      // statistics of the 32-row slab = this wave's rows (layout and arithmetic of ff_fused_kernel / mimo_gemm_ext's colstats:
      // mean, then the sum of squared deviations from it; fixed order)
      auto store_tile = [&](auto nt_c, const f32x4& v0, const f32x4& v1) {
        constexpr int nt = decltype(nt_c)::value;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rO32, o_off + 64u * (unsigned)nt, 0, MIMO_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rO32, o_off + 64u * (unsigned)nt, o_mi, MIMO_ST_AUX);
        {   // (without a statistics buffer the descriptor is empty: the stores are dropped, no branch in the tile loop)
          f32x4 sm = v0 + v1;
#pragma unroll
          for (int r = 0; r < 4; ++r) sm[r] = row16_sum(sm[r]);
          const f32x4 mean_c = sm * (1.0f / 32.0f);
          const f32x4 d0 = v0 - mean_c, d1 = v1 - mean_c;
          f32x4 qv = d0 * d0 + d1 * d1;
#pragma unroll
          for (int r = 0; r < 4; ++r) qv[r] = row16_sum(qv[r]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mean_c), rCS, cs_off + 64u * (unsigned)nt, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), rCS, cs_off + 64u * (unsigned)nt, (unsigned)(C * 4), 0);
        }
      };
      if constexpr (TRICKLE) {
        // a finished column-tile group: + x, out (8 stores, 16 with the statistics)
        auto finish = [&](auto gq_c) {
          static_for<0, 4>([&](auto k_c) {
            constexpr int nt = (decltype(k_c)::value < 2 ? 0 : 10) + 2 * decltype(gq_c)::value + (decltype(k_c)::value & 1);
            store_tile(ICf<nt>{}, acc2[nt][0] + opr[nt][0], acc2[nt][1] + opr[nt][1]);
          });
        };
        // x group q is fetched under tile q - 1 (group 0 at the drain) and used when tile q + 1 is nearly through: at most three
        // groups in flight.  Counted waits, behind the pieces of the tile about to be read: tile 1 — x groups 0, 1: 16 loads;
        // tiles 2, 3, 4 — 8 loads + the 16 stores of a finished group.
        proj(ICf<0>{}, ICf<0>{}, [&](auto k_c) { load_part(opr, rX, x_off, x_mi, ICf<1>{}, k_c); }, no_tail, ICf<9>{}); ++t;
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 24);
        proj(ICf<1>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, x_mi, ICf<2>{}, k_c); }, [&]() { finish(ICf<0>{}); }, ICf<9>{}); ++t;
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 25);
        proj(ICf<2>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, x_mi, ICf<3>{}, k_c); }, [&]() { finish(ICf<1>{}); }, ICf<9>{}); ++t;
        asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 26);
        proj(ICf<3>{}, ICf<1>{}, [&](auto k_c) { load_part(opr, rX, x_off, x_mi, ICf<4>{}, k_c); }, [&]() { finish(ICf<2>{}); }, ICf<9>{}); ++t;
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
          static_for<0, 4>([&](auto k_c) { load_part(opr, rRn, pinned(r_off0), r_mi0, ICf<0>{}, k_c); });
        }
        // an empty asm that READS the next operand: the compiler places its own counted wait for the 20 loads in front of it (it
        // knows the 40 operations it issued behind them) and from here on treats them as landed
        asm volatile("" :: "v"(__builtin_bit_cast(u32x4, fa_next[0][0])), "v"(__builtin_bit_cast(u32x4, fa_next[0][1])), "v"(__builtin_bit_cast(u32x4, fa_next[0][2])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[0][3])), "v"(__builtin_bit_cast(u32x4, fa_next[0][4])), "v"(__builtin_bit_cast(u32x4, fa_next[0][5])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[0][6])), "v"(__builtin_bit_cast(u32x4, fa_next[0][7])), "v"(__builtin_bit_cast(u32x4, fa_next[0][8])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[0][9])), "v"(__builtin_bit_cast(u32x4, fa_next[1][0])), "v"(__builtin_bit_cast(u32x4, fa_next[1][1])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[1][2])), "v"(__builtin_bit_cast(u32x4, fa_next[1][3])), "v"(__builtin_bit_cast(u32x4, fa_next[1][4])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[1][5])), "v"(__builtin_bit_cast(u32x4, fa_next[1][6])), "v"(__builtin_bit_cast(u32x4, fa_next[1][7])),
                     "v"(__builtin_bit_cast(u32x4, fa_next[1][8])), "v"(__builtin_bit_cast(u32x4, fa_next[1][9])) : "memory");
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) fa[mi][ks] = fa_next[mi][ks];
      } else {
        proj(ICf<0>{}, ICf<0>{}, no_hook, no_tail, ICf<0>{});
        ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 24); proj(ICf<1>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 25); proj(ICf<2>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 26); proj(ICf<3>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); FF_TRACE(g, tr, 27); proj(ICf<4>{}, ICf<1>{}, no_hook, no_tail, ICf<0>{}); ++t;
        static_assert(NTAIL == 5, "projection tiles are spelled out");
        FF_TRACE(g, tr, 28);
        static_for<0, 20>([&](auto nt_c) { store_tile(nt_c, acc2[decltype(nt_c)::value][0], acc2[decltype(nt_c)::value][1]); });
      }
    }
  }
  FF_TRACE(g, tr, 30);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
  FF_TRACE(g, tr, 31);
}

template <int DT>
void ff4_launch_dt(const FFArgs& g, int mode, unsigned grid, hipStream_t st) {
  if (mode == 2) hipLaunchKernelGGL((ff4_kernel<DT, 2>), dim3(grid), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((ff4_kernel<DT, 0>), dim3(grid), dim3(256), 0, st, g);
}

}  // namespace

// called by ff_launch (ff_fused.hip) for MODE 0 / 2
void mimo_ff4_launch(int dtype, const void* args, int mode, unsigned grid, void* stream) {
  const FFArgs& g = *static_cast<const FFArgs*>(args);
  if (dtype == MIMO_F16) ff4_launch_dt<MIMO_F16>(g, mode, grid, (hipStream_t)stream);
  else ff4_launch_dt<MIMO_BF16>(g, mode, grid, (hipStream_t)stream);
}
