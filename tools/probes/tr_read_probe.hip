// Probe of ds_read_b64_tr_b16 (gfx950): which LDS halfs does each lane receive?  LDS is filled with half index = value,
// every lane passes address = base + 8 * lane (linear) and, in a second pass, a row-major [kv][40] V-tile style address.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) {
    addr = 8u * lane;  // linear: lane l points at halfs [4 l, 4 l + 4)
  } else {
    // V tile row-major [kv][40 halfs]; group g = lane >> 4, m = lane & 15: row kvbase(g) + (m >> 2), d chunk 4 (m & 3)
    const int g = lane >> 4, m = lane & 15;
    const int kvbase = 16 * (g & 1) + 4 * (g >> 1);
    addr = (unsigned)(((kvbase + (m >> 2)) * 40 + 4 * (m & 3)) * 2);
  }
  addr += (unsigned)(size_t)(__attribute__((address_space(3))) void*)&lds[0];
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = (uint16_t)(v.x & 0xffff);
  out[lane * 4 + 1] = (uint16_t)(v.x >> 16);
  out[lane * 4 + 2] = (uint16_t)(v.y & 0xffff);
  out[lane * 4 + 3] = (uint16_t)(v.y >> 16);
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        if (mode == 0) printf(" %4d", h[l * 4 + j]);
        else printf(" (kv %2d, d %2d)", h[l * 4 + j] / 40, h[l * 4 + j] % 40);
      }
      printf("\n");
    }
  }
  return 0;
}
