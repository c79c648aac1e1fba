// What MFMA rate does THIS chip sustain with nothing but MFMAs in flight?  (round-5 verdict item 6: "write the ceiling down")
// Every wave runs a loop of independent v_mfma_f32_16x16x32_f16 (the instruction of the GEMM / conv kernels) or
// v_mfma_f32_32x32x16_f16 (attention) on register operands: no LDS, no memory, no VALU besides the loop counter.  Operands are
// random halfs (mode 'r') or zeros (mode 'z'): the chip clocks to its power budget, and zeros draw less (MI355X_MICROARCH.md
// "DVFS give-back").  Reports TFLOP/s from the wall clock (hipEvents) and the effective shader clock = s_memtime ticks / wall.
// Build here (cross-compiles), run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_ceiling_probe.hip -o tools/_ab/mfma_ceiling_probe && tools/_ab/mfma_ceiling_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>  // 0: 16x16x32 (16 independent accumulators of 4 regs), 1: 32x32x16 (8 accumulators of 16 regs)
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* seed, float* sink, unsigned long long* cyc, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = seed[(tid * 8 + i) & 4095]; b[i] = seed[(tid * 8 + 4 + i) & 4095]; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  float s = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[(i >> 2) & 1]), acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678f) sink[tid & 1023] = s;  // keeps the accumulators alive
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// VALU issue rates (the softmax of attn40 is one v_exp_f32 + half a v_cvt_pk + half a v_or per score): cycles per wave64
// instruction of a SIMD, from s_memtime over a loop of 8 independent chains.  OP 0: v_exp_f32, 1: v_fma_f32, 2: v_cvt_pk_f16_f32,
// 3: v_exp_f32 and v_fma_f32 alternating (do they share the pipe?)
template <int OP>
__global__ __launch_bounds__(512, 2) void valu_loop(float* sink, unsigned long long* cyc, int iters) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = -0.001f * (float)(threadIdx.x + i + 1);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
      else if (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x[i]));
      else { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[(i + 4) & 7])); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// How many VALU instructions hide behind one MFMA of the SAME wave?  Loop body: one v_mfma_f32_32x32x16_f16 (4 independent
// accumulators round-robin) followed by NV independent VALU instructions (OP 0: v_exp_f32, 1: v_fma_f32, 2: v_cvt_pk_f16_f32).
// Reports cycles per loop body (= per MFMA) for one wave; the MFMA alone is 32 cycles of its SIMD's matrix pipe.
template <int OP, int NV>
__global__ __launch_bounds__(512, 2) void mix_loop(const uint4* seed, float* sink, unsigned long long* cyc, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const f16x8 a = __builtin_bit_cast(f16x8, seed[tid & 4095]), b = __builtin_bit_cast(f16x8, seed[(tid + 77) & 4095]);
  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  float x[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) x[i] = -0.001f * (float)(threadIdx.x + i + 1);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define MIX_VALU()                                                                 \
  _Pragma("unroll") for (int i = 0; i < NV; ++i) {                                  \
    if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));                     \
    else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));        \
    else asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x[i]));                  \
  }
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
    MIX_VALU()
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
    MIX_VALU()
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
    MIX_VALU()
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
    MIX_VALU()
#undef MIX_VALU
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += x[i];
  s += acc0[0] + acc1[15] + acc2[3] + acc3[7];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int NV>
static void mix_one(int cus, const uint4* seed, float* sink, unsigned long long* cyc, const char* name) {
  const int iters = 4000;
  double res[2];
  for (int wps = 1; wps <= 2; ++wps) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((mix_loop<OP, NV>), dim3(cus), dim3(256 * wps), 0, 0, seed, sink, cyc, iters);
      hipDeviceSynchronize();
    }
    unsigned long long c[2048];
    hipMemcpy(c, cyc, cus * 8, hipMemcpyDeviceToHost);
    double cm = 0;
    for (int i = 0; i < cus; ++i) cm += (double)c[i];
    res[wps - 1] = cm / cus / ((double)iters * 4);
  }
  printf("1 MFMA 32x32x16 + %2d %-17s: %6.1f cycles per MFMA with one wave per SIMD, %6.1f with two (each wave; per SIMD: %.1f)\n", NV, name, res[0], res[1],
         res[1] / 2);
}

static void mix_rates(int cus, const uint4* seed, float* sink, unsigned long long* cyc) {
  printf("# in-wave overlap: cycles per (MFMA + NV VALU instructions of the same wave); the bare MFMA is 32 matrix-pipe cycles\n");
  mix_one<1, 0>(cus, seed, sink, cyc, "(none)");
  mix_one<1, 2>(cus, seed, sink, cyc, "v_fma_f32");
  mix_one<1, 4>(cus, seed, sink, cyc, "v_fma_f32");
  mix_one<1, 6>(cus, seed, sink, cyc, "v_fma_f32");
  mix_one<1, 8>(cus, seed, sink, cyc, "v_fma_f32");
  mix_one<1, 12>(cus, seed, sink, cyc, "v_fma_f32");
  mix_one<0, 2>(cus, seed, sink, cyc, "v_exp_f32");
  mix_one<0, 4>(cus, seed, sink, cyc, "v_exp_f32");
  mix_one<0, 6>(cus, seed, sink, cyc, "v_exp_f32");
  mix_one<0, 8>(cus, seed, sink, cyc, "v_exp_f32");
  mix_one<0, 12>(cus, seed, sink, cyc, "v_exp_f32");
  mix_one<2, 4>(cus, seed, sink, cyc, "v_cvt_pk_f16_f32");
  mix_one<2, 8>(cus, seed, sink, cyc, "v_cvt_pk_f16_f32");
}

// Do the matrix pipe and the VALU of one SIMD run concurrently when the work comes from DIFFERENT waves?  Block = 8 waves; waves
// w and w + 4 share a SIMD.  Waves 0-3 run an MFMA-only loop (mode bit 0), waves 4-7 a VALU-only loop of v_exp_f32 / v_fma_f32
// (mode bit 1); wall time by hipEvents for MFMA waves alone, VALU waves alone, both.  Overlap: both ~= max; none: both ~= sum.
template <int OP>
__global__ __launch_bounds__(512, 2) void role_loop(const uint4* seed, float* sink, unsigned long long* cyc, int iters_m, int iters_v, int mode) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (!(mode & 1)) return;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const f16x8 a = __builtin_bit_cast(f16x8, seed[tid & 4095]), b = __builtin_bit_cast(f16x8, seed[(tid + 77) & 4095]);
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
    for (int it = 0; it < iters_m; ++it) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b));
    }
    s = acc0[0] + acc1[15] + acc2[3] + acc3[7];
  } else {
    if (!(mode & 2)) return;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = -0.001f * (float)(threadIdx.x + i + 1);
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if ((threadIdx.x & 255) == 0) cyc[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;   // wave 0: MFMA role, wave 4: VALU role
}

template <int OP>
static void role_one(int cus, const uint4* seed, float* sink, unsigned long long* cyc, const char* name, int iters_m, int iters_v) {
  float ms[4] = {0, 0, 0, 0};
  double cm[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 1; mode <= 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipMemset(cyc, 0, cus * 16);
      hipLaunchKernelGGL((role_loop<OP>), dim3(cus), dim3(512), 0, 0, seed, sink, cyc, iters_m, iters_v, mode);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
      unsigned long long c[4096];
      hipMemcpy(c, cyc, cus * 16, hipMemcpyDeviceToHost);
      cm[mode] = cv[mode] = 0;
      for (int i = 0; i < cus; ++i) { cm[mode] += (double)c[2 * i] / cus; cv[mode] += (double)c[2 * i + 1] / cus; }
    }
  printf("MFMA waves alone %7.3f ms | %s waves alone %7.3f ms | both %7.3f ms  (sum %.3f, max %.3f)\n", ms[1], name, ms[2], ms[3], ms[1] + ms[2],
         ms[1] > ms[2] ? ms[1] : ms[2]);
  printf("    shader cycles of a wave (s_memtime): MFMA role alone %.3e, with the partner %.3e | VALU role alone %.3e, with the partner %.3e\n",
         cm[1], cm[3], cv[2], cv[3]);
}

static void role_rates(int cus, const uint4* seed, float* sink, unsigned long long* cyc) {
  printf("# cross-wave concurrency of the matrix pipe and the VALU on one SIMD (waves w: MFMA 32x32x16 only, w + 4: VALU only)\n");
  role_one<1>(cus, seed, sink, cyc, "v_fma_f32", 40000, 40000);   // 160k MFMAs x 32 cyc vs 320k fma x 8 cyc per wave
  role_one<0>(cus, seed, sink, cyc, "v_exp_f32", 40000, 27000);
}

static void valu_rates(int cus, float* sink, unsigned long long* cyc) {
  const int iters = 20000;
  const char* names[4] = {"v_exp_f32", "v_fma_f32", "v_cvt_pk_f16_f32", "v_exp_f32 + v_fma_f32 (pair)"};
  printf("# VALU: s_memtime cycles per wave64 instruction and SIMD (1 | 2 waves per SIMD issuing)\n");
  for (int op = 0; op < 4; ++op)
    for (int wps = 1; wps <= 2; ++wps) {
      const int threads = 256 * wps;
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) hipLaunchKernelGGL(valu_loop<0>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
        else if (op == 1) hipLaunchKernelGGL(valu_loop<1>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
        else if (op == 2) hipLaunchKernelGGL(valu_loop<2>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
        else hipLaunchKernelGGL(valu_loop<3>, dim3(cus), dim3(threads), 0, 0, sink, cyc, iters);
        hipDeviceSynchronize();
      }
      unsigned long long c[2048];
      hipMemcpy(c, cyc, cus * 8, hipMemcpyDeviceToHost);
      double cm = 0;
      for (int i = 0; i < cus; ++i) cm += (double)c[i];
      cm /= cus;
      const double per_wave_instr = (double)iters * 8 * (op == 3 ? 2 : 1);
      printf("%-30s %d wave/SIMD: %.2f cycles per instruction of one wave, %.2f per instruction of the SIMD\n", names[op], wps,
             cm / per_wave_instr, cm / (per_wave_instr * wps));
    }
}

int main(int argc, char** argv) {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  uint4* seed; float* sink; unsigned long long* cyc;
  hipMalloc(&seed, 4096 * 16); hipMalloc(&sink, 4096); hipMalloc(&cyc, 8 * cus * 4);
  uint16_t* h = (uint16_t*)malloc(4096 * 16);
  const int iters = argc > 1 ? atoi(argv[1]) : 40000;
  printf("# %d CUs; %d loop iterations; per (shape, waves per SIMD, data): TFLOP/s by wall clock, shader clock = s_memtime ticks / wall\n", cus, iters);
  for (int data = 0; data < 2; ++data) {
    srand(7);
    for (int i = 0; i < 4096 * 8; ++i) {
      // random halfs in [-2, 2): sign, exponent 13..16 (2^-2 .. 2^1), random mantissa; or zeros
      const uint16_t r = (uint16_t)(((rand() & 1) << 15) | ((13 + (rand() & 3)) << 10) | (rand() & 0x3ff));
      h[i] = data == 0 ? r : 0;
    }
    hipMemcpy(seed, h, 4096 * 16, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape)
      for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps, blocks = cus;  // one block per CU, wps waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {  // the last repetition is reported (clocks settled)
          hipEventRecord(e0, 0);
          if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, seed, sink, cyc, iters);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, seed, sink, cyc, iters);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
        }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2048];
        hipMemcpy(c, cyc, blocks * 8, hipMemcpyDeviceToHost);
        double cm = 0;
        for (int i = 0; i < blocks; ++i) cm += (double)c[i];
        cm /= blocks;
        const double flop_per_mfma = shape == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
        const double n_mfma = (double)iters * (shape == 0 ? 16 : 8) * blocks * (threads / 64);
        // s_memtime counts at a constant 100 MHz on gfx9 (REFCLK); report both the raw tick rate and cycles per MFMA assuming shader ticks
        printf("%s  %d wave/SIMD  %s  %8.1f TFLOP/s  wall %7.3f ms  memtime ticks %.3e (%.3f GHz if shader cycles)\n",
               shape == 0 ? "16x16x32" : "32x32x16", wps, data == 0 ? "random" : "zeros ", n_mfma * flop_per_mfma / (ms * 1e-3) / 1e12, ms,
               cm, cm / (ms * 1e-3) / 1e9);
      }
  }
  valu_rates(cus, sink, cyc);
  mix_rates(cus, seed, sink, cyc);
  role_rates(cus, seed, sink, cyc);
  return 0;
}
