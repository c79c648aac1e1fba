// thinconv.hip.h — internal interface between gemm_conv.hip (the mimo_conv2d launcher) and thinconv.hip (kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimo_thin {

struct Args {
  const uint16_t* in;   // half [n, Hin, Win, Cin]
  const uint16_t* W;    // half [Cout, ldw]: tap-major packed rows (mimo_amd.packing.pack_conv), 9 Cin columns used
  void* out;            // half or fp32 [n, Hout, Wout, Cout]
  const float* bias;    // [Cout] or null
  int n, Hin, Win, Cin, Hout, Wout, Cout, ksize, stride, pad_t, pad_l;
  int64_t ldw;
  float out_scale;
  unsigned flags;       // MIMO_EPI_SILU | MIMO_EPI_OUT_F32
  unsigned in_bytes, w_bytes, out_bytes;
};

// true when the direct kernel implements this layer (a function of the layer only): 3x3 with Cin in {8, 16} and Cout <= 128 or
// Cin = 32 and Cout <= 32; 1x1 stride 1 with Cin in {128, 320} and Cout <= 48
bool supported(int Cin, int Cout, int ksize, int stride, int64_t in_bytes, int64_t out_bytes);
int launch(int dtype, const Args& a, int cus, hipStream_t st);

}  // namespace mimo_thin
