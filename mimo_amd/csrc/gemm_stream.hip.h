// gemm_stream.hip.h — internal interface between gemm_conv.hip (launcher) and gemm_stream.hip (kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimo_stream {

struct Args {
  const uint16_t* A;     // half [M, lda], K = 320 columns used
  const uint16_t* W;     // half [N, K] (GEGLU: packed [16 value | 16 gate] row blocks)
  void* out;             // half [M, ldo] (GEGLU: N / 2 columns)
  const float* bias;     // [N] or null
  int64_t lda, ldo, M;
  int N;
  int geglu;
  unsigned long long* dbg;  // tune build trace buffer or null
  int ablate;               // tune build timing experiments (results are wrong): 1 no B-fragment reads, 2 no stores, 3 no DMA, 4 no MFMA
};

// true when the streaming kernel implements this problem (K == 320, N % 64 == 0, N <= 4096, and — unless any_m — enough
// rows to fill the chip)
bool supported(int64_t M, int N, int K, bool any_m);
int launch(int dtype, const Args& a, int cus, hipStream_t st);

}  // namespace mimo_stream
