// What does a feed-forward stream position of the fused block tail cost on 32x32x16 MFMAs instead of 16x16x32?  (round-5 / 6 verdict
// item 3.)  The position of ff4_kernel (ff_tail4.hip) as a skeleton: ONE wave per SIMD, 30 fenced segments of 64 matrix-pipe cycles,
// each with two ds_read_b128 W fragments (fetched two segments ahead), NV independent VALU instructions (the GEGLU stage), one
// LDS-DMA piece of three instructions in 15 of the segments, `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` per position.
//   FORM 0: 4 x v_mfma_f32_16x16x32_f16 per segment (independent accumulators)          — what ff4_kernel issues today
//   FORM 1: 2 x v_mfma_f32_32x32x16_f16 per segment, both into the SAME accumulator     — the natural FF1 chain (one accumulator per pair)
//   FORM 2: 2 x v_mfma_f32_32x32x16_f16 per segment into two alternating accumulators
// Reports s_memtime cycles per position (matrix pipe alone: 1 920).  Build here, run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/ff_position_probe.hip -o tools/_ab/ff_position_probe && tools/_ab/ff_position_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int I> struct IC { static constexpr int value = I; };
template <int A, int B, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (A < B) { f(IC<A>{}); static_for<A + 1, B>(f); }
}

constexpr int STAGE = 61440;   // W1 tile 40 KB + W2 slice 20 KB, two stages: 120 KB of LDS as in the kernel

template <int FORM, int NV, int DMA>
__global__ __launch_bounds__(256, 1) void position_loop(const uint4* wsrc, float* sink, unsigned long long* cyc, int npos) {
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * STAGE / 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned pr = __builtin_amdgcn_readfirstlane((unsigned)tid >> 6);
  for (int n = tid; n < 2 * STAGE / 16; n += 256) smem[n] = wsrc[n & 4095];
  __syncthreads();
  const unsigned smem_base = (unsigned)(size_t)(lds_ptr_t)&smem[0];
  i32x4 rW;
  { const uint64_t a = reinterpret_cast<uint64_t>(wsrc); rW.x = (int)(uint32_t)a; rW.y = (int)((uint32_t)(a >> 32) & 0xffffu); rW.z = 65536; rW.w = 0x00020000; }
  const unsigned voff = (unsigned)lane * 16u;
  i32x4 junk = {0, 0, 0, 0};
  auto dma = [&](i32x4& junk_, unsigned voff_, unsigned sbase, auto sconst_c, unsigned dbase, auto dconst_c) {
    const i32x4 r = {__builtin_amdgcn_readfirstlane(rW.x), __builtin_amdgcn_readfirstlane(rW.y), __builtin_amdgcn_readfirstlane(rW.z),
                     __builtin_amdgcn_readfirstlane(rW.w)};
    unsigned soff;
    if constexpr (DMA == 1)
      asm volatile("s_add_u32 m0, %4, %5\n\ts_add_u32 %0, %3, %6\n\tbuffer_load_dwordx4 %1, %2, %0 offen lds"
                   : "=&s"(soff)
                   : "v"(voff_), "s"(r), "s"(__builtin_amdgcn_readfirstlane(sbase)), "s"(__builtin_amdgcn_readfirstlane(dbase)),
                     "i"(decltype(dconst_c)::value), "i"(decltype(sconst_c)::value)
                   : "memory", "m0", "scc");
    else if constexpr (DMA == 2)   // the load alone: m0 / soffset set once per position, the piece index in the 12-bit offset field
      asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds"
                   :: "v"(voff_), "s"(r), "s"(__builtin_amdgcn_readfirstlane(sbase)), "i"(decltype(dconst_c)::value / 4096 % 4 * 1024) : "memory");
    else if constexpr (DMA == 3)   // the two scalar additions alone
      asm volatile("s_add_u32 m0, %2, %3\n\ts_add_u32 %0, %1, %4"
                   : "=&s"(soff)
                   : "s"(__builtin_amdgcn_readfirstlane(sbase)), "s"(__builtin_amdgcn_readfirstlane(dbase)),
                     "i"(decltype(dconst_c)::value), "i"(decltype(sconst_c)::value)
                   : "memory", "m0", "scc");
    else if constexpr (DMA == 4)   // a plain 16-byte load into registers nobody waits for until the position barrier (no LDS destination)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                   : "=v"(junk_) : "v"(voff_), "s"(r), "s"(__builtin_amdgcn_readfirstlane(sbase)), "i"(decltype(dconst_c)::value / 4096 % 4 * 1024) : "memory");
  };
  // fragment addresses: FORM 0 — 16 rows x 4 chunks; FORM 1 / 2 — 32 rows x 2 chunks; row pitch 128 B, chunk XOR-swizzled by the row
  const unsigned li = lane & 15, lg = lane >> 4, i32 = lane & 31, h = lane >> 5;
  const unsigned q16 = li * 8u + (lg ^ (li & 7u));
  const unsigned q32 = i32 * 8u + (h ^ (i32 & 7u));
  uint4 fa[20];
#pragma unroll
  for (int k = 0; k < 20; ++k) fa[k] = wsrc[(tid * 20 + k) & 4095];
  f32x4 acc16[8];
  f32x16 acc32[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc16[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[k][r] = 0.f;
  float x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = 0.001f * (float)(lane + k + 1);
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int t = 0; t < npos; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned sq = (unsigned)(t & 1) * (unsigned)(STAGE / 16);
    const unsigned sb = __builtin_amdgcn_readfirstlane(0u);
    const unsigned db = __builtin_amdgcn_readfirstlane(smem_base + (unsigned)((t + 1) & 1) * (unsigned)STAGE + pr * 1024u);
    if constexpr (DMA == 2) asm volatile("s_mov_b32 m0, %0" :: "s"(db) : "m0", "memory");
    uint4 fr[3][2];
    auto frag_load = [&](auto seg_c, uint4 (&dst)[2]) {
      constexpr int seg = decltype(seg_c)::value;
      const unsigned q = sq + (FORM == 0 ? q16 : q32) + (unsigned)((seg % 5) * 512);
      dst[0] = smem[q + (unsigned)(seg / 5 & 1) * 128u];
      dst[1] = smem[q + 256u + (unsigned)(seg / 10) * 128u];
    };
    frag_load(IC<0>{}, fr[0]);
    frag_load(IC<1>{}, fr[1]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 30>([&](auto seg_c) {
      constexpr int seg = decltype(seg_c)::value;
      if constexpr (seg + 2 < 30) frag_load(IC<seg + 2>{}, fr[(seg + 2) % 3]);
      if constexpr (DMA != 0) {
        if constexpr (seg < 10) dma(junk, voff, sb, IC<seg * 1024>{}, db, IC<seg * 4096>{});
        else if constexpr (seg < 20 && (seg & 1) != 0) dma(junk, voff, sb, IC<(10 + seg / 2) * 1024>{}, db, IC<40960 + ((seg - 10) / 2) * 4096>{});
      }
      const uint4(&src)[2] = fr[seg % 3];
      if constexpr (FORM == 0) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc16[(seg & 1) * 4 + 2 * n + mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(f16x8, src[n]), __builtin_bit_cast(f16x8, fa[(seg % 10) * 2 + mi]), acc16[(seg & 1) * 4 + 2 * n + mi], 0, 0, 0);
      } else {
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc32[FORM == 1 ? 0 : n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, src[n]),
                                                                            __builtin_bit_cast(f16x8, fa[(seg % 10) * 2 + n]),
                                                                            acc32[FORM == 1 ? 0 : n], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[k]));
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += x[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += acc16[k][0] + acc16[k][3];
  s += acc32[0][0] + acc32[1][15];
  if (s == 12345.678f) sink[tid] = s + (float)junk.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int FORM, int NV, int DMA>
static void run(int cus, const uint4* w, float* sink, unsigned long long* cyc) {
  const int npos = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((position_loop<FORM, NV, DMA>), dim3(cus), dim3(256), 0, 0, w, sink, cyc, npos);
    hipDeviceSynchronize();
  }
  unsigned long long c[2048];
  hipMemcpy(c, cyc, cus * 8, hipMemcpyDeviceToHost);
  double cm = 0;
  for (int i = 0; i < cus; ++i) cm += (double)c[i];
  static const char* names[3] = {"4 x 16x16x32 per segment", "2 x 32x32x16, one accumulator", "2 x 32x32x16, two accumulators"};
  printf("%-32s  %2d VALU per segment  DMA %d : %7.0f cycles per position\n", names[FORM], NV, DMA, cm / cus / npos);
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  uint4* w; float* sink; unsigned long long* cyc;
  hipMalloc(&w, 65536); hipMalloc(&sink, 4096); hipMalloc(&cyc, 8 * cus);
  uint16_t* hbuf = (uint16_t*)malloc(65536);
  srand(7);
  for (int i = 0; i < 32768; ++i) hbuf[i] = (uint16_t)(((rand() & 1) << 15) | ((11 + (rand() & 3)) << 10) | (rand() & 0x3ff));
  hipMemcpy(w, hbuf, 65536, hipMemcpyHostToDevice);
  printf("# %d CUs, one block of 4 waves per CU; 30 segments of 64 matrix-pipe cycles per position (1 920)\n", cus);
  printf("# DMA 0: none | 1: the kernel's three-instruction piece | 2: the load alone (m0 once per position) | 3: the two s_add alone | 4: a register load\n");
  run<0, 8, 0>(cus, w, sink, cyc); run<0, 8, 1>(cus, w, sink, cyc); run<0, 8, 2>(cus, w, sink, cyc); run<0, 8, 3>(cus, w, sink, cyc); run<0, 8, 4>(cus, w, sink, cyc);
  run<1, 8, 0>(cus, w, sink, cyc); run<1, 8, 1>(cus, w, sink, cyc); run<1, 8, 2>(cus, w, sink, cyc); run<1, 8, 3>(cus, w, sink, cyc); run<1, 8, 4>(cus, w, sink, cyc);
  run<0, 12, 0>(cus, w, sink, cyc); run<0, 12, 1>(cus, w, sink, cyc); run<0, 12, 2>(cus, w, sink, cyc);
  run<1, 12, 0>(cus, w, sink, cyc); run<1, 12, 1>(cus, w, sink, cyc); run<1, 12, 2>(cus, w, sink, cyc);
  run<0, 0, 0>(cus, w, sink, cyc); run<1, 0, 0>(cus, w, sink, cyc); run<0, 4, 0>(cus, w, sink, cyc); run<1, 4, 0>(cus, w, sink, cyc);
  return 0;
}
