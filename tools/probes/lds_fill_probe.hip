// How fast can a CU pull bytes, and does the LDS-DMA path (buffer_load ... lds) cap it?  The fused tail / head, the whole-row
// N = 640 tiles and the small-M token GEMMs all sit at ~12 B/clk per CU of operand fill (NOTEBOOK round 5 §9); this probe
// separates the candidates: the instruction form (LDS-DMA b128 | LDS-DMA b32 | plain b128 into VGPRs), where the bytes
// come from (a window every block shares, sized to sit in the TCP / the XCD's L2 / the MALL, or distinct windows = an HBM
// stream), how much is in flight per wave (DEPTH KB) and how many waves pull (4 | 8 per block, 1 | 2 blocks per CU).
// Every wave moves 1 KB pieces of its block's window round-robin with a counted s_waitcnt (DEPTH always in flight).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_fill_probe.hip -o tools/_ab/lds_fill_probe && tools/_ab/lds_fill_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: LDS-DMA b128 (1 KB per instruction), 1: b128 into VGPRs (1 KB), 2: LDS-DMA b32 (256 B per instruction)
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill(const char* base, unsigned window, unsigned long long block_stride, int passes, unsigned* sink) {
  extern __shared__ char smem[];
  const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const unsigned long long a = (unsigned long long)(base + blockIdx.x * block_stride);
  const i32x4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                   __builtin_amdgcn_readfirstlane((int)window), 0x00020000};
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, (int)window, 0x00020000);
  constexpr unsigned PIECE = MODE == 2 ? 256u : 1024u;
  const unsigned voff = lane * (MODE == 2 ? 4u : 16u);
  const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * (unsigned)DEPTH * 1024u;
  const unsigned npieces = window / PIECE;
  unsigned acc = 0;
  for (int p = 0; p < passes; ++p) {
    if (MODE == 1) {
      const unsigned total = npieces < nw * DEPTH ? nw * DEPTH : npieces;   // a window smaller than one round is re-read
      const unsigned mask = npieces < nw * DEPTH ? npieces - 1u : 0xffffffffu;   // (small windows are powers of two)
      for (unsigned i = wave * DEPTH; i + DEPTH <= total; i += nw * DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, ((i + k) & mask) * PIECE, 0);
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) acc ^= v[k].x ^ v[k].w;
      }
    } else {
      unsigned slot = 0;
      for (unsigned i = wave; i < npieces; i += nw) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * PIECE);
        const unsigned soff = __builtin_amdgcn_readfirstlane(i * PIECE);
        if (MODE == 0)
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(%4)"
                       :: "v"(voff), "s"(r), "s"(soff), "s"(dst), "n"(DEPTH - 1) : "memory", "m0");
        else
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(%4)"
                       :: "v"(voff), "s"(r), "s"(soff), "s"(dst), "n"(DEPTH - 1) : "memory", "m0");
        slot = slot + 1 == (unsigned)(DEPTH * 1024u / PIECE) ? 0u : slot + 1;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != 1) acc = ((volatile unsigned*)smem)[threadIdx.x];
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* what, const char* buf, size_t bufsz, unsigned window, bool shared, int blocks, int threads, double target_bytes_per_block) {
  const unsigned long long stride = shared ? 0ull : window;
  if (!shared && (size_t)blocks * window > bufsz) { printf("skip %s\n", what); return; }
  int passes = (int)(target_bytes_per_block / window);
  if (passes < 1) passes = 1;
  unsigned* sink;
  CHECK(hipMalloc(&sink, 4096));
  const size_t lds = (size_t)(threads / 64) * DEPTH * 1024;
  CHECK(hipFuncSetAttribute((const void*)fill<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  fill<MODE, DEPTH><<<blocks, threads, lds>>>(buf, window, stride, 1, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  fill<MODE, DEPTH><<<blocks, threads, lds>>>(buf, window, stride, passes, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const unsigned round = (unsigned)(threads / 64) * DEPTH * 1024u;
  const double bytes = (double)blocks * passes * (MODE == 1 && window < round ? round : window);
  const int cus = blocks < 256 ? blocks : 256;
  printf("%-44s %s depth %2d  %4d x %3d thr  window %9u B %-8s  %7.2f TB/s  %6.1f B/ns per CU\n", what,
         MODE == 0 ? "lds-dma b128" : MODE == 1 ? "vgpr    b128" : "lds-dma b32 ", DEPTH, blocks, threads, window, shared ? "shared" : "distinct",
         bytes / ms * 1e-9, bytes / (ms * 1e6) / cus);
  CHECK(hipFree(sink));
}

int main() {
  const size_t bufsz = 4ull << 30;
  char* buf;
  CHECK(hipMalloc(&buf, bufsz));
  CHECK(hipMemset(buf, 1, bufsz));
  const double per_block = 48e6;
  printf("# bytes / ns per CU: divide by the shader clock in GHz (2.0-2.4) for B/clk per CU\n");
  struct W { const char* name; unsigned window; bool shared; };
  const W ws[] = {{"shared 32 KB (TCP-resident)", 32u << 10, true}, {"shared 1 MB (L2-resident)", 1u << 20, true},
                  {"shared 3 MB (a fused tail's weights, L2)", 3u << 20, true}, {"shared 12 MB (> L2, MALL)", 12u << 20, true},
                  {"shared 64 MB (MALL)", 64u << 20, true}, {"distinct 4 MB per block (HBM stream)", 4u << 20, false}};
  for (const W& w : ws) {
    run<0, 16>(w.name, buf, bufsz, w.window, w.shared, 256, 256, per_block);
    run<0, 16>(w.name, buf, bufsz, w.window, w.shared, 256, 512, per_block);
    run<0, 16>(w.name, buf, bufsz, w.window, w.shared, 512, 256, per_block);
    run<0, 4>(w.name, buf, bufsz, w.window, w.shared, 256, 512, per_block);
    run<0, 8>(w.name, buf, bufsz, w.window, w.shared, 256, 512, per_block);
    run<1, 16>(w.name, buf, bufsz, w.window, w.shared, 256, 256, per_block);
    run<1, 16>(w.name, buf, bufsz, w.window, w.shared, 256, 512, per_block);
    run<1, 16>(w.name, buf, bufsz, w.window, w.shared, 512, 512, per_block);
    run<2, 16>(w.name, buf, bufsz, w.window, w.shared, 256, 512, per_block);
  }
  return 0;
}
