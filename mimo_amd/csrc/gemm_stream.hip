// gemm_stream.hip — "A in registers, W streamed" GEMM for the level-0 token linears of the UNets (gfx950).
//
//   out[M, N] = epi(A[M, K] @ W[N, K]^T + bias),   K = 320, M = 10^5 rows, N = 960 (QKV) | 2560 (GEGLU FF1)
//
// Why a second GEMM kernel: with K = 320 the tiled kernel of gemm_conv.hip restarts its pipeline every five K-tiles
// (prologue latency + a 256 x 256 epilogue with the matrix pipe idle), re-reads the A tile once per N-tile and moves
// (BM + BN) x 128 B through LDS per K-tile: tools/gemm_trace.py measures 4200 cycles per K-tile against 2048 of MFMA
// work for the GEGLU shape.  Here a block owns a panel of 256 rows for ALL N: each wave keeps its 32 x 320 slice of A
// in registers (80 VGPRs, loaded once per panel in MFMA operand layout), so LDS carries only W, and W is ONE continuous
// stream of 64-row x 320-column tiles (40 KB) through a 3-deep ring filled by LDS-DMA two tiles ahead — no per-N-tile
// restart, the epilogue of tile j runs under the DMAs of tiles j + 1, j + 2.
//
// Block: 8 waves, wave w owns rows [32 w, 32 w + 32) of the panel and ALL 64 columns of a step (16 x 16 x 32 MFMAs,
// swapped: lane holds 4 consecutive columns of one row): 80 VGPRs of A, 32 of accumulators.  Per step and wave 40
// ds_read_b128 (B fragments) feed 80 MFMAs: 320 KB of LDS reads + 40 KB of DMA writes per step = 1400 cycles at
// 256 B/clk against 2560 MFMA cycles per SIMD.
// W tile image in LDS: row r at r * 640 B, 16-byte chunk c of a row stored at chunk (c & ~7) | ((c & 7) ^ (r & 7))
// (the swizzle is applied on the DMA's source address), so the 16 lanes of a ds_read_b128 group hit 16 distinct
// 16-byte bank groups.
// Output: every 16-row MFMA tile is transposed through a wave-private LDS patch so that stores are full 128-byte lines.
//
// Memory-operation accounting (per wave, in issue order): ... DMA(t) | stores(t-2) | DMA(t+1) | stores(t-1) | ...
// so "tile t has landed" is a COUNTED s_waitcnt vmcnt at the top of step t (vmcnt retires in order on gfx9).
// The bias lives in LDS (ds_read, not a vector-memory load) so that the epilogue adds no loads to that queue.
#include "common.hip.h"
#include "gemm_stream.hip.h"

namespace mimo_stream {
namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifdef MIMO_TUNE
// two tracers: thread 0 (wave 0, stores after its MFMAs) fills entries [0, 2000), thread 256 (wave 4, stores first) [2000, 4000)
#define STREAM_TRACE(g, idx, tag)                                                                                  \
  do {                                                                                                             \
    if ((g).dbg && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (idx) < 2000u)                                   \
      (g).dbg[(threadIdx.x >> 8) * 2000u + (idx)++] =                                                              \
          ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull);              \
  } while (0)
#define STREAM_TRACE_REAL(g, idx, tag)                                                                             \
  do {                                                                                                             \
    if ((g).dbg && blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (idx) < 2000u)                                   \
      (g).dbg[(threadIdx.x >> 8) * 2000u + (idx)++] =                                                              \
          ((unsigned long long)(tag) << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull);          \
  } while (0)
#define STREAM_ABLATE(g, n) ((g).ablate == (n))
#else
#define STREAM_ABLATE(g, n) false
#define STREAM_TRACE(g, idx, tag) do { (void)(idx); } while (0)
#define STREAM_TRACE_REAL(g, idx, tag) do { (void)(idx); } while (0)
#endif

constexpr int MAXN = 4096;  // bias image in LDS
constexpr int NW = 8;        // waves per block (two per SIMD; 12 = three per SIMD leaves the compiler no registers to prefetch
                             // B fragments under its 168-register cap: every MFMA pair then waits for its own LDS read, 2.5x slower)
constexpr int BM_ROWS = 32 * NW;

template <int V>
struct IC {
  static constexpr int value = V;
};

template <int DT, int KS, bool GEGLU>
__global__ __launch_bounds__(64 * NW, NW / 4) void gemm_stream_kernel(const Args g) {
  constexpr int K = 32 * KS;
  constexpr int ROWB = K * 2;                // bytes of one W row
  constexpr int BN = 64, BM = BM_ROWS, NST = 3;
  constexpr int TILE_B = BN * ROWB;          // 40 KB at K = 320
  constexpr int NDMA = NW == 8 ? 5 : 4;      // 1-KB DMAs per tile of the waves that load (waves 0 .. NLOADW - 1)
  constexpr int NLOADW = TILE_B / (NDMA * 1024);
  constexpr int MT = 2, NR = 4;              // wave tile: 32 rows x 64 packed columns
  constexpr int OC = GEGLU ? 32 : 64;        // output columns of a wave per step
  constexpr int NSTORE = GEGLU ? 2 : 4;      // 16-byte stores per wave per step
  constexpr int PITCH = OC * 2 + 16;         // bytes per row of the wave's transposition patch (16 rows)
  constexpr int PATCH_B = 16 * PITCH;
  constexpr unsigned OOBA = 0x80000000u;
  static_assert(TILE_B % (NDMA * 1024) == 0 && NLOADW <= NW && K % 64 == 0, "tile must be whole DMAs, rows whole 128-byte groups");
  static_assert(NDMA + 2 * NSTORE < 64, "vmcnt is 6 bits");
  static_assert(NST * TILE_B + MAXN * 4 + NW * PATCH_B <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) uint4 smem[(NST * TILE_B + MAXN * 4 + NW * PATCH_B) / 16];  // ONE LDS object

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, li = lane & 15;
  const int NT = g.N / BN;
  const unsigned npanels = (unsigned)((g.M + BM - 1) / BM);
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  float* const bias_lds = reinterpret_cast<float*>(&smem[NST * TILE_B / 16]);
  constexpr unsigned PATCH_Q = (NST * TILE_B + MAXN * 4) / 16;  // 16-byte index of patch 0

  for (int n = tid; n < g.N; n += 64 * NW) bias_lds[n] = g.bias ? g.bias[n] : 0.f;

  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const i32x4 rW = make_rsrc(g.W, (unsigned)g.N * (unsigned)ROWB);

  // ---- W stream: DMA d = wave * NDMA + i of a tile fills LDS bytes [d * 1024, d * 1024 + 1024) of the stage ----
  unsigned w_voff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const unsigned b = (wave_u * NDMA + i) * 1024u + (unsigned)lane * 16u;
    const unsigned row = b / (unsigned)ROWB, pc = (b % (unsigned)ROWB) >> 4;
    const unsigned lc = (pc & ~7u) | ((pc & 7u) ^ (row & 7u));
    w_voff[i] = row * (unsigned)ROWB + lc * 16u;
  }
  auto dma = [&](unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rW), "s"(__builtin_amdgcn_readfirstlane(soff)), "s"(lds_dst) : "memory", "m0");
  };
  // ring position `t` (global tile counter of this block) -> N-tile t % NT, stage t % NST; both kept incrementally
  const unsigned my_panels = blockIdx.x < npanels ? (npanels - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
  const unsigned total = my_panels * (unsigned)NT;
  unsigned ld_t = 0, ld_j = 0, ld_s = 0;
  const bool loader = wave_u < (unsigned)NLOADW;  // 40 pieces per tile over the first NLOADW waves
  auto issue_next = [&]() {
    ++ld_t;
    if (!loader) return;
    const bool live = ld_t <= total;
    const unsigned dst = smem_base + ld_s * (unsigned)TILE_B + wave_u * (NDMA * 1024u);
    const unsigned soff = ld_j * (unsigned)TILE_B;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma(live && !STREAM_ABLATE(g, 3) ? w_voff[i] : OOBA, soff, dst + i * 1024u);
    ld_j = ld_j + 1 == (unsigned)NT ? 0u : ld_j + 1;
    ld_s = ld_s + 1 == (unsigned)NST ? 0u : ld_s + 1;
  };

  // ---- B-fragment addresses: tile row ni*16 + li, logical chunk 4*ks + lg (every wave reads the whole W tile) ----
  const unsigned brow = (unsigned)li * (unsigned)ROWB;
  const unsigned bq0 = (brow >> 4) + (unsigned)(lg ^ (li & 7));        // even ks: chunks 0..3 of the 128-byte group
  const unsigned bq1 = (brow >> 4) + (unsigned)((4 + lg) ^ (li & 7));  // odd ks: chunks 4..7

  unsigned tr = 0;
  STREAM_TRACE_REAL(g, tr, 0xfe);
  STREAM_TRACE(g, tr, 1);
  issue_next();
  issue_next();
  __syncthreads();  // bias image complete (the compiler drains its own loads; the DMAs are invisible to it)

  // ---- epilogue of one 32 x 64 wave tile.  The MFMA result has lanes along ROWS (lane li = row, 4 consecutive columns
  // per lane): stored as it is, every lane is its own 8- or 16-byte write request and the CU's one-request-per-clock
  // store path, not HBM, bounds a short-K GEMM.  So each 16-row MFMA tile goes through a wave-private LDS patch and
  // comes back with lanes along COLUMNS: 8 lanes x 16 B = one full 128-byte line of one output row per request
  // group (GEGLU: 4 lanes = 64 B).  Wave-private: DS operations of a wave execute in order, no barrier needed. ----
  const int n_out = GEGLU ? g.N / 2 : g.N;
  const unsigned patch_q = PATCH_Q + wave_u * (PATCH_B / 16);
  constexpr int LPR = OC * 2 / 16;                    // lanes per output row on the way out (8 | 4)
  constexpr int RPI = 64 / LPR;                       // rows per store instruction (8 | 16)
  const unsigned o_lane = (unsigned)(((int64_t)(wave_u * 32 + lane / LPR) * g.ldo) * 2 + (lane % LPR) * 16);
  const unsigned o_rpi = (unsigned)(RPI * g.ldo * 2);
  auto epilogue = [&](f32x4 (&acc)[NR][MT], const __amdgpu_buffer_rsrc_t& rO, int j) {
    f32x4 bv[NR];
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(bias_lds + j * BN + ni * 16 + 4 * lg);
    const unsigned oj = o_lane + (unsigned)(j * OC * 2);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      char* const pw = reinterpret_cast<char*>(&smem[patch_q]) + li * PITCH + lg * 8;
      if (GEGLU) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {  // (value, gate) = MFMA tiles (2 pr, 2 pr + 1) -> output columns 16 pr ..
          const f32x4 v = acc[2 * pr][mi] + bv[2 * pr], gt = acc[2 * pr + 1][mi] + bv[2 * pr + 1];
          u32x2 o;
          const f32x4 h = v * gelu_erf_4(gt);
          o.x = pack2<DT>(h[0], h[1]);
          o.y = pack2<DT>(h[2], h[3]);
          *reinterpret_cast<u32x2*>(pw + pr * 32) = o;
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NR; ++ni) {
          const f32x4 v = acc[ni][mi] + bv[ni];
          u32x2 o;
          o.x = pack2<DT>(v[0], v[1]); o.y = pack2<DT>(v[2], v[3]);
          *reinterpret_cast<u32x2*>(pw + ni * 32) = o;
        }
      }
      const char* const pr_ = reinterpret_cast<const char*>(&smem[patch_q]) + (lane / LPR) * PITCH + (lane % LPR) * 16;
#pragma unroll
      for (int k = 0; k < 16 / RPI; ++k) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(pr_ + k * RPI * PITCH);
        __builtin_amdgcn_raw_buffer_store_b128(o, rO, STREAM_ABLATE(g, 2) ? OOBA : oj + (unsigned)(mi * 16 / RPI + k) * o_rpi, 0, 0);
      }
    }
  };

  // The two waves of a SIMD (w, w + 4) run half a step out of phase: waves 0-3 compute tile t and then store it, waves 4-7
  // first store their tile t - 1 and then compute tile t, so on every SIMD one wave's epilogue runs under the other
  // wave's MFMAs.  (Both orders issue the same memory operations per step: the counted wait holds.)
  const bool late_store = (wave_u & 4u) != 0u;
  f32x4 acc[NR][MT];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __amdgpu_buffer_rsrc_t rO_prev = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, 0, 0x00020000);  // zero extent: stores are dropped
  int j_prev = 0;

  unsigned cs = 0;  // stage of the tile being computed
  for (unsigned panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const int64_t M0 = (int64_t)panel * BM;
    const int64_t rows_valid = (g.M - M0) < (int64_t)BM ? (g.M - M0) : (int64_t)BM;
    // ---- this wave's 32 x K slice of A, straight into MFMA operand layout ----
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)const_cast<uint16_t*>(g.A + M0 * g.lda), 0, (int)(((rows_valid - 1) * g.lda + K) * 2), 0x00020000);
    const unsigned a_off = (unsigned)(((int64_t)(wave_u * 32 + li) * g.lda + lg * 8) * 2);
    const unsigned a_mi = (unsigned)(16 * g.lda * 2);
    uint4 fa[MT][KS];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        fa[mi][ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, a_off + mi * a_mi + ks * 64, 0, 0));
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(
        (char*)g.out + M0 * g.ldo * 2, 0, (int)(((rows_valid - 1) * g.ldo + n_out) * 2), 0x00020000);
    STREAM_TRACE(g, tr, 8);

    for (int j = 0; j < NT; ++j) {
      STREAM_TRACE(g, tr, 2);
      // this wave's part of the tile has landed: memory operations issued after DMA(t) are
      //   early waves: stores(t-2) DMA(t+1) stores(t-1);   late waves: stores(t-2) DMA(t+1)
      if (loader) {
        if (late_store) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + 2 * NSTORE) : "memory");
      }  // (waves without pieces of the tile wait for nothing: the barrier below covers them)
      STREAM_TRACE(g, tr, 7);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // all parts landed; stage (cs + 2) % 3 is free
      STREAM_TRACE(g, tr, 3);
      if (late_store) {
        epilogue(acc, rO_prev, j_prev);
        rO_prev = rO;
        j_prev = j;
        STREAM_TRACE(g, tr, 10);
        issue_next();  // the DMA issue (about 100 cycles a piece) of one wave also runs under the other wave's MFMAs
        STREAM_TRACE(g, tr, 9);
      }
      const unsigned sq = cs * (unsigned)(TILE_B / 16);  // 16-byte index of the stage
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // B fragments run PF k-steps ahead of the MFMAs that consume them
      constexpr int PF = 2;
      uint4 fb[KS][NR];
      auto ldb = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        const unsigned q = sq + ((ks & 1) ? bq1 : bq0) + (unsigned)((ks >> 1) * 8);  // + (ks >> 1) * 128 bytes
        if (STREAM_ABLATE(g, 1)) {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) fb[ks][ni] = fa[ni & 1][ks];
        } else {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) fb[ks][ni] = smem[q + ni * (16 * ROWB / 16)];  // 16 rows apart
        }
      };
      auto kstep = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        if constexpr (ks + PF < KS) ldb(IC<ks + PF>{});
        if (STREAM_ABLATE(g, 4)) {
#pragma unroll
          for (int ni = 0; ni < NR; ++ni) acc[ni][0][0] += __builtin_bit_cast(float, fb[ks][ni].x);
          return;
        }
#pragma unroll
        for (int ni = 0; ni < NR; ++ni)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = HT<DT>::mfma16(fb[ks][ni], fa[mi][ks], acc[ni][mi]);
      };
      ldb(IC<0>{}); ldb(IC<1>{});
      static_assert(KS == 10, "k-steps are spelled out");
      kstep(IC<0>{}); kstep(IC<1>{}); kstep(IC<2>{}); kstep(IC<3>{}); kstep(IC<4>{});
      kstep(IC<5>{}); kstep(IC<6>{}); kstep(IC<7>{}); kstep(IC<8>{}); kstep(IC<9>{});
      STREAM_TRACE(g, tr, 4);
      STREAM_TRACE(g, tr, 5);
      if (!late_store) {
        issue_next();
        STREAM_TRACE(g, tr, 9);
        epilogue(acc, rO, j);
      }
      STREAM_TRACE(g, tr, 6);
      cs = cs + 1 == (unsigned)NST ? 0u : cs + 1;
    }
  }
  if (late_store) epilogue(acc, rO_prev, j_prev);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS
  STREAM_TRACE_REAL(g, tr, 0xff);
}

}  // namespace

// any_m: the caller needs the kernel choice to be a function of (N, K) only — batch-invariant runs (MIMO_EPI_NO_SPLITK) must
// route a b = 1 unit and the b = 2 launch of the same window through the SAME kernel whatever their row counts
bool supported(int64_t M, int N, int K, bool any_m) {
  return K == 320 && N % 64 == 0 && N >= 640 && N <= MAXN && M > 0 && (any_m || M >= (int64_t)BM_ROWS * 64);
}

int launch(int dtype, const Args& a, int cus, hipStream_t st) {
  if (!supported(a.M, a.N, 320, true)) return MIMO_EINVAL;
  const int64_t npanels = (a.M + BM_ROWS - 1) / BM_ROWS;
  if (npanels > 0x7fffffff) return MIMO_EINVAL;
  const unsigned grid = (unsigned)(npanels < cus ? npanels : cus);
  if (dtype == MIMO_F16) {
    if (a.geglu) hipLaunchKernelGGL((gemm_stream_kernel<MIMO_F16, 10, true>), dim3(grid), dim3(64 * NW), 0, st, a);
    else hipLaunchKernelGGL((gemm_stream_kernel<MIMO_F16, 10, false>), dim3(grid), dim3(64 * NW), 0, st, a);
  } else if (dtype == MIMO_BF16) {
    if (a.geglu) hipLaunchKernelGGL((gemm_stream_kernel<MIMO_BF16, 10, true>), dim3(grid), dim3(64 * NW), 0, st, a);
    else hipLaunchKernelGGL((gemm_stream_kernel<MIMO_BF16, 10, false>), dim3(grid), dim3(64 * NW), 0, st, a);
  } else {
    return MIMO_EDTYPE;
  }
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

}  // namespace mimo_stream
