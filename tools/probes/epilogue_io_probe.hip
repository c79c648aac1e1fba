// Probe: does the access pattern of the GEMM epilogues (MFMA "swapped" layout: lane li = row, 16 B = 4 fp32 columns
// per lane, so the 64 lanes of one load/store instruction touch 16 rows x 4 separate 16-byte pieces) cap the
// read-modify-write rate of an fp32 [M, 320] tensor, compared with the same bytes moved as whole rows
// (consecutive lanes = consecutive 16-byte chunks)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/epi_probe tools/probes/epilogue_io_probe.hip && /tmp/epi_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave per 16-row x 80-column patch, 8 waves per block = 128 rows x 80 columns; grid covers [M/128][4]
template <int MODE>
__global__ __launch_bounds__(512) void rmw(const float* __restrict__ src, float* __restrict__ dst, int M, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  const int col0 = blockIdx.y * 80;
  if (MODE == 0) {  // epilogue pattern
    f32x4 v[5];
#pragma unroll
    for (int ni = 0; ni < 5; ++ni) v[ni] = *reinterpret_cast<const f32x4*>(src + (row0 + li) * N + col0 + ni * 16 + 4 * lg);
#pragma unroll
    for (int ni = 0; ni < 5; ++ni) *reinterpret_cast<f32x4*>(dst + (row0 + li) * N + col0 + ni * 16 + 4 * lg) = v[ni] + 1.f;
  } else {  // row-linear pattern: chunk q = k * 64 + lane of the 16 x 80 patch (20 chunks per row)
    f32x4 v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int q = k * 64 + lane, r = q / 20, c = q % 20;
      v[k] = *reinterpret_cast<const f32x4*>(src + (row0 + r) * N + col0 + c * 4);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int q = k * 64 + lane, r = q / 20, c = q % 20;
      *reinterpret_cast<f32x4*>(dst + (row0 + r) * N + col0 + c * 4) = v[k] + 1.f;
    }
  }
}

int main() {
  const int M = 196608, N = 320;
  float *a, *b;
  hipMalloc(&a, (size_t)M * N * 4);
  hipMalloc(&b, (size_t)M * N * 4);
  hipMemset(a, 0, (size_t)M * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) {
        if (mode == 0) rmw<0><<<dim3(M / 128, 4), 512>>>(a, b, M, N);
        else rmw<1><<<dim3(M / 128, 4), 512>>>(a, b, M, N);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.1f us per pass, %.2f TB/s (read + write of %d x %d fp32)\n", mode ? "row-linear 16-B chunks" : "MFMA epilogue pattern ",
                      ms / 20 * 1e3, 2.0 * M * N * 4 / (ms / 20 * 1e-3) / 1e12, M, N);
    }
  }
  return 0;
}
