// gemm_conv.hip — the MFMA workhorse of the denoising path: one LDS-tiled kernel that is
// either a dense GEMM  out[M,N] = A[M,K].W[N,K]^T  or a channels-last implicit-GEMM
// convolution (3x3 / 1x1, stride 1|2, optional nearest-upsample gather, optional extra 1x1
// tap over a second tensor = fused ResBlock shortcut), with a fused epilogue
// (bias, per-batch bias = time embedding / collapsed cross-attention, SiLU, GEGLU,
// residual add, fp32|half store).
//
// Reference ops replaced: see include/mimo_hip.h (mimo_gemm / mimo_conv2d).
//
// Tiling (gfx950): block = 2*WM waves laid out WM x 2; block tile (64*WM) x (32*NR) x 64;
// each wave owns 64 x (16*NR) as 4 x NR MFMA 16x16x32 tiles, fp32 accumulators in VGPRs.
// Operands go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`): hardware range checking
// turns padding taps / ragged edges into zero fill, the XOR swizzle that makes the
// ds_read_b128 fragment reads bank-conflict-free is applied on the SOURCE address (the DMA
// writes lane-linear), and an NSTAGE-deep LDS ring keeps NSTAGE-1 K-tiles in flight under
// the MFMAs with ONE barrier and one counted `s_waitcnt vmcnt` per K-tile.
//   WM = 2, NSTAGE = 2 (128-row tile, 72-80 KB LDS, 2 blocks/CU): small / skinny problems
//   WM = 4, NSTAGE = 3 (256-row tile, 144-159 KB LDS, 1 block/CU): large-M problems
// The MFMA is issued "swapped" (W fragment as the A operand) so that every lane ends up
// with 4 consecutive output columns of one row -> 16-byte epilogue loads/stores.
#include <stdlib.h>

#include "common.cuh"

namespace {

constexpr int BK = 64;

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* A2;
  const uint16_t* W;
  void* out;
  const float* bias;
  const float* img_bias;
  const void* res;
  int64_t lda, ldw, ldo, ldr, M, rows_per_img, ldib;
  unsigned a_bytes, a2_bytes, w_bytes;  // buffer-descriptor extents (each operand < 4 GiB)
  int N, K;
  float out_scale;
  unsigned flags;
  int tiles_n;
  // conv geometry
  int Hin, Win, Cin, Hout, Wout, ks, stride, pad_t, pad_l, Hup, Wup, Cin2;
  float sh, sw;
  int chunks1, chunks2, nkt;
};

template <int V>
struct IC {
  static constexpr int value = V;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0: dense GEMM; 1: convolution gather; 2: convolution gather through a nearest-neighbour upsampling
template <int DT, int NR, int MODE, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void gemm_kernel(const GemmArgs g) {
  constexpr bool CONV = MODE != 0;
  constexpr int THREADS = 64 * WM * WN;
  constexpr int BM = 64 * WM;
  constexpr int BN = 16 * NR * WN;
  constexpr int PASS = THREADS / 8;                   // tile rows covered by one DMA pass of the whole block
  constexpr int NAJ = BM / PASS;                      // A passes
  constexpr int NBJ = (BN + PASS - 1) / PASS;         // B passes
  constexpr int BNA = NBJ * PASS > BN ? BN + 8 : BN;  // + 8 dummy rows that absorb the surplus (all-zero) DMAs
  constexpr int LOADS = NAJ + NBJ;                    // DMAs per thread per K-tile
  static_assert(BM % PASS == 0, "A tile must be a whole number of DMA passes");
  constexpr int STAGE = (BM + BNA) * 8;               // 16-byte units per ring slot: [A tile | B tile]
  // ONE LDS object (a second __shared__ array would make hipcc drain vmcnt before fragment reads)
  __shared__ __attribute__((aligned(16))) uint4 smem[NSTAGE * STAGE];

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

  // issue the LOADS DMAs of K-tile kt into ring slot `slot` (kt >= nkt: all-zero DMAs keep the counts uniform)
  auto load_tile = [&](int kt, int slot) {
    // offsets are computed on (uniform) branches; the DMAs themselves are issued once, after the join
    unsigned offA[NAJ], kw = 0;
    bool cok = false;
    bool main_tap = true;
    const bool live = kt < g.nkt;
    if (CONV) {
      const int ntap_tiles = g.ks * g.ks * g.chunks1;
      main_tap = kt < ntap_tiles;
      if (main_tap) {
        const int tap = kt / g.chunks1;
        const int c0 = (kt - tap * g.chunks1) * BK;
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

  f32x4 acc[NR][4];
#pragma unroll
  for (int ni = 0; ni < NR; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](auto slot_c) {
    constexpr int S = decltype(slot_c)::value;
    const uint4* sa = &smem[S * STAGE];
    const uint4* sb = &smem[S * STAGE + BM * 8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ch = (4 * s + lg) ^ (li & 7);
      uint4 fa[4], fb[NR];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = sa[(wm * 64 + mi * 16 + li) * 8 + ch];
#pragma unroll
      for (int ni = 0; ni < NR; ++ni) fb[ni] = sb[(wn * 16 * NR + ni * 16 + li) * 8 + ch];
#pragma unroll
      for (int ni = 0; ni < NR; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          // swapped: D[row = n-in-tile = 4*lg + r][col = m-in-tile = li]
          acc[ni][mi] = HT<DT>::mfma16(fb[ni], fa[mi], acc[ni][mi]);
    }
  };

  // One K-tile: wait until everything but the newest NSTAGE-2 tiles of THIS wave has landed, barrier (all
  // waves' parts landed AND every wave is done reading the slot about to be recycled), refill that slot with
  // tile kt + NSTAGE - 1, then run the MFMAs of tile kt while the DMAs fly.
  auto step = [&](auto slot_c, int kt) {
    constexpr int S = decltype(slot_c)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LOADS) : "memory");
    __syncthreads();
    if (!(g.flags & 0x10000u)) load_tile(kt + NSTAGE - 1, (S + NSTAGE - 1) % NSTAGE);  // (ablation: no DMA)
    if (!(g.flags & 0x20000u)) compute(slot_c);                                        // (ablation: no MFMA)
  };

  const int nkt = g.nkt;
#pragma unroll
  for (int i = 0; i < NSTAGE - 1; ++i) load_tile(i, i);
  for (int kt = 0; kt < nkt; kt += NSTAGE) {
    step(IC<0>{}, kt);
    if (kt + 1 < nkt) step(IC<1 % NSTAGE>{}, kt + 1);
    if (NSTAGE > 2 && kt + 2 < nkt) step(IC<2 % NSTAGE>{}, kt + 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing all-zero DMAs must not outlive the block's LDS

  // ---- epilogue: lane holds out[m = .. + li][n = .. + 4*lg + r], r = 0..3 ----
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32;
  const bool res_f32 = g.flags & MIMO_EPI_RES_F32;
  const bool do_silu = g.flags & MIMO_EPI_SILU;
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int64_t m = M0 + wm * 64 + mi * 16 + li;
    if (m >= g.M) continue;
    const int64_t img = g.img_bias ? m / g.rows_per_img : 0;
#pragma unroll
    for (int ni = 0; ni < NR; ++ni) {
      if (geglu && (ni & 1)) continue;
      const int n = N0 + wn * 16 * NR + ni * 16 + 4 * lg;
      if (n >= g.N) continue;
      float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
      if (g.bias) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (g.img_bias) {
        const float4 b = *reinterpret_cast<const float4*>(g.img_bias + img * g.ldib + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      int no = n;
      if (geglu) {
        // gate tile = the next 16 packed columns; identical lane mapping
        const int ni1 = (ni + 1 < NR) ? ni + 1 : ni;
        float gt[4] = {acc[ni1][mi][0], acc[ni1][mi][1], acc[ni1][mi][2], acc[ni1][mi][3]};
        if (g.bias) {
          const float4 b = *reinterpret_cast<const float4*>(g.bias + n + 16);
          gt[0] += b.x; gt[1] += b.y; gt[2] += b.z; gt[3] += b.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_f(gt[r]);
        no = (N0 + wn * 16 * NR + ni * 16) / 2 + 4 * lg;
      }
      if (do_silu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      if (g.res) {
        if (res_f32) {
          const float4 r4 = *reinterpret_cast<const float4*>((const float*)g.res + m * g.ldr + no);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        } else {
          const uint2 r2 = *reinterpret_cast<const uint2*>((const uint16_t*)g.res + m * g.ldr + no);
          v[0] += HT<DT>::to_f((uint16_t)(r2.x & 0xffffu));
          v[1] += HT<DT>::to_f((uint16_t)(r2.x >> 16));
          v[2] += HT<DT>::to_f((uint16_t)(r2.y & 0xffffu));
          v[3] += HT<DT>::to_f((uint16_t)(r2.y >> 16));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= g.out_scale;
      if (out_f32) {
        *reinterpret_cast<float4*>((float*)g.out + m * g.ldo + no) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        uint2 o;
        o.x = pack2<DT>(v[0], v[1]);
        o.y = pack2<DT>(v[2], v[3]);
        *reinterpret_cast<uint2*>((uint16_t*)g.out + m * g.ldo + no) = o;
      }
    }
  }
}

template <int DT, int MODE, int NR>
int launch_nr(GemmArgs& g, hipStream_t st) {
  // Tile configurations (WM x WN waves, ring depth):
  //   S  2x2, 2 stages: 128 x 32NR tile, 2 blocks/CU          — small / skinny problems
  //   L  4x2, 3 stages: 256 x 32NR tile, 1 block/CU
  //   XL 4x4, 2 stages: 256 x 64NR tile (N = 320 in ONE tile), 16 waves, 1 block/CU — halves the global->LDS
  //      bytes per MAC; the L2->LDS path (~18 TB/s measured) and not the MFMAs bounds the S tile
  // MIMO_GEMM_CFG=1|2|3 forces S|L|XL (tuning knob for A/B runs); MIMO_GEMM_ABLATE is for timing experiments.
  static const int forced = getenv("MIMO_GEMM_CFG") ? atoi(getenv("MIMO_GEMM_CFG")) : 0;
  static const int ablate = getenv("MIMO_GEMM_ABLATE") ? atoi(getenv("MIMO_GEMM_ABLATE")) : 0;
  if (ablate == 1) g.flags |= 0x10000u;
  if (ablate == 2) g.flags |= 0x20000u;
  const int64_t m256 = (g.M + 255) / 256;
  const int tn_s = (g.N + 32 * NR - 1) / (32 * NR), tn_xl = (g.N + 64 * NR - 1) / (64 * NR);
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
  int cfg = forced;
  if (cfg == 0) {
    if (g.N <= 32 * NR) cfg = (m256 * tn_s >= 160) ? 2 : 1;  // one S-width tile covers N: a 64NR-wide XL tile would idle
    else cfg = (m256 * tn_xl >= 160) ? 3 : 1;                 // measured crossover (tools/microbench.py)
  }
  if (cfg == 3) {
    g.tiles_n = tn_xl;
    const int64_t nwg = m256 * tn_xl;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, 4, 4, 2>), dim3((unsigned)nwg), dim3(1024), 0, st, g);
  } else if (cfg == 2) {
    g.tiles_n = tn_s;
    const int64_t nwg = m256 * tn_s;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, 4, 2, 3>), dim3((unsigned)nwg), dim3(512), 0, st, g);
  } else {
    g.tiles_n = tn_s;
    const int64_t nwg = ((g.M + 127) / 128) * tn_s;
    if (nwg <= 0 || nwg > 0x7fffffff) return MIMO_EINVAL;
    hipLaunchKernelGGL((gemm_kernel<DT, NR, MODE, 2, 2, 2>), dim3((unsigned)nwg), dim3(256), 0, st, g);
  }
  (void)geglu;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

template <int DT, int MODE>
int launch(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  const bool geglu = g.flags & MIMO_EPI_GEGLU;
  // NR = 5 (BN = 160) divides every SD1.5 width (320/640/960/1280/1920/2560); NR = 4 otherwise
  if (!geglu && (g.N % 160 == 0)) return launch_nr<DT, MODE, 5>(g, st);
  return launch_nr<DT, MODE, 4>(g, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int mimo_gemm(int dtype, const void* A, int64_t lda, const void* W, void* out, int64_t ldo,
                         int64_t M, int N, int K, const float* bias, const float* img_bias,
                         int64_t img_bias_ld, int64_t rows_per_img, const void* residual, int64_t ldr,
                         float out_scale, unsigned flags, void* stream) {
  if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0) return MIMO_EINVAL;
  if ((K & 7) || (lda & 7) || (N & 3) || (ldo & 3) || !aligned16(A) || !aligned16(W) || !aligned16(out))
    return MIMO_EINVAL;
  if ((flags & MIMO_EPI_GEGLU) && (N & 31)) return MIMO_EINVAL;
  if (residual && (ldr & 3)) return MIMO_EINVAL;
  if (img_bias && (rows_per_img <= 0 || (img_bias_ld & 3))) return MIMO_EINVAL;
  GemmArgs g{};
  g.A = (const uint16_t*)A; g.A2 = nullptr; g.W = (const uint16_t*)W; g.out = out;
  g.bias = bias; g.img_bias = img_bias; g.res = residual;
  g.lda = lda; g.ldw = K; g.ldo = ldo; g.ldr = ldr; g.M = M; g.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
  g.ldib = img_bias_ld > 0 ? img_bias_ld : N;
  g.N = N; g.K = K; g.out_scale = out_scale; g.flags = flags;
  g.nkt = (K + BK - 1) / BK;
  const int64_t ab = ((M - 1) * lda + K) * 2, wb = (int64_t)N * K * 2;
  if (ab >= 0xFFFFFFF0LL || wb >= 0xFFFFFFF0LL) return MIMO_EINVAL;  // operands are addressed with 32-bit offsets
  g.a_bytes = (unsigned)ab; g.a2_bytes = 0; g.w_bytes = (unsigned)wb;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16) return launch<MIMO_F16, 0>(g, st);
  if (dtype == MIMO_BF16) return launch<MIMO_BF16, 0>(g, st);
  return MIMO_EDTYPE;
}

extern "C" int mimo_conv2d(int dtype, const void* in, const void* in2, const void* W, void* out,
                           const mimo_conv_params* p, const float* bias, const float* img_bias,
                           const void* residual, float out_scale, unsigned flags, void* stream) {
  if (!in || !W || !out || !p) return MIMO_EINVAL;
  if (p->n <= 0 || p->Cin <= 0 || (p->Cin & 7) || (p->Cout & 3) || p->Cout <= 0) return MIMO_EINVAL;
  if (!(p->ksize == 1 || p->ksize == 3) || !(p->stride == 1 || p->stride == 2)) return MIMO_EINVAL;
  if (p->Cin2 < 0 || (p->Cin2 & 7) || (p->Cin2 > 0 && !in2)) return MIMO_EINVAL;
  if ((p->Hup > 0) != (p->Wup > 0)) return MIMO_EINVAL;
  if (flags & MIMO_EPI_GEGLU) return MIMO_EINVAL;
  if (!aligned16(in) || !aligned16(W) || !aligned16(out) || (in2 && !aligned16(in2))) return MIMO_EINVAL;
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
  g.Hin = p->Hin; g.Win = p->Win; g.Cin = p->Cin; g.Hout = p->Hout; g.Wout = p->Wout;
  g.ks = p->ksize; g.stride = p->stride; g.pad_t = p->pad_t; g.pad_l = p->pad_l;
  g.Hup = p->Hup; g.Wup = p->Wup; g.Cin2 = p->Cin2;
  g.sh = p->Hup > 0 ? (float)p->Hin / (float)p->Hup : 1.f;
  g.sw = p->Wup > 0 ? (float)p->Win / (float)p->Wup : 1.f;
  g.chunks1 = (p->Cin + BK - 1) / BK;
  g.chunks2 = (p->Cin2 + BK - 1) / BK;
  g.nkt = p->ksize * p->ksize * g.chunks1 + g.chunks2;
  {
    const int64_t ab = (int64_t)p->n * p->Hin * p->Win * p->Cin * 2, a2b = g.M * p->Cin2 * 2, wb = (int64_t)g.N * g.K * 2;
    if (ab >= 0xFFFFFFF0LL || a2b >= 0xFFFFFFF0LL || wb >= 0xFFFFFFF0LL) return MIMO_EINVAL;
    g.a_bytes = (unsigned)ab; g.a2_bytes = (unsigned)a2b; g.w_bytes = (unsigned)wb;
  }
  const bool ups = p->Hup > 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16) return ups ? launch<MIMO_F16, 2>(g, st) : launch<MIMO_F16, 1>(g, st);
  if (dtype == MIMO_BF16) return ups ? launch<MIMO_BF16, 2>(g, st) : launch<MIMO_BF16, 1>(g, st);
  return MIMO_EDTYPE;
}

extern "C" int mimo_version(void) { return 2; }
