// thinconv.hip — direct 3x3 convolution for the thin-input layers (gfx950): Cin in {8, 16, 32}, Cout <= 128 — and, with
// one tap, the thin-N GEMM (N <= 48, K = 128 | 320) that is the first half of a thin-OUTPUT convolution (mimo_conv3x3_tapsum).
//
//   replaces nn.Conv2d(3|16|32 -> 16|32, 3, stride 1|2) + SiLU of PoseGuider.forward (src/models/pose_guider.py:47-57) and the
//   VAE encoder's conv_in (diffusers AutoencoderKL.encoder.conv_in, called from pipeline_..._roiclip.py:430) behind
//   mimo_conv2d: the launcher of gemm_conv.hip routes a call here when mimo_thin::supported() says so.
//
// Why a second convolution kernel: the implicit-GEMM kernel walks K as (tap, 64-channel chunk) tiles and N as 80..320-column
// tiles.  With 8 or 16 input channels a K-tile is 12..25 % data, with 16 output channels an N-tile is 5..20 % data: these
// layers ran at 9..70 TFLOP/s, 10-20x over the time their bytes take (profiles/r3_vae_bound_512.txt).  They are HBM-bound by
// construction (at 512 x 512 x 16 channels the output is as large as the input and the weights are a few KB), so the kernel
// is organised around the bytes, not the MFMAs:
//   * the whole weight [Cout][9 Cin] sits in LDS (<= 37 KB), loaded once per block;
//   * a wave owns strips of 16 consecutive output pixels of one row and ALL output channels; K = 9 Cin runs tap-major in
//     k-steps of 32 = 4 chunks of 8 channels, so the A fragment of the 16x16x32 MFMA (lane = pixel li, chunk lg) is ONE
//     16-byte load per lane straight from the image (chunk q = 4 s + lg -> tap q / (Cin/8), channels 8 (q % (Cin/8)) ..):
//     the im2col matrix exists in registers only, the 9 shifted reads of a pixel hit L1/L2, padding pixels and the K tail
//     are buffer-range zeros;
//   * MFMAs are issued swapped (weights as the A operand): lane (li, lg) ends with output channels 4 lg .. 4 lg + 3 of pixel
//     li, i.e. 8 (half) or 16 (fp32) contiguous bytes per lane and whole pixel rows per 16-lane group on the way out;
//   * no barrier after the weight load: waves are independent, many blocks per CU hide the load latency.
// Epilogue: bias, optional SiLU, out_scale, half or fp32 output (what these layers use; anything else stays on the
// implicit-GEMM kernel).
#include "common.hip.h"
#include "thinconv.hip.h"

namespace mimo_thin {
namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V>
struct IC {
  static constexpr int value = V;
};

template <int DT, int CIN, int NT, int TAPS = 9>
__global__ __launch_bounds__(256) void thin_conv_kernel(const Args g) {
  constexpr int CPT = CIN / 8;            // 16-byte chunks per tap
  constexpr int NCH = TAPS * CPT;         // chunks of a weight row (TAPS = 1: a 1x1 convolution = a GEMM with a thin N)
  constexpr int KS = (NCH + 3) / 4;       // k-steps of 32
  constexpr int PITCH = KS * 4 + 1;       // 16-byte units per weight row in LDS (odd: the 16 rows of a fragment read spread over the banks)
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) uint4 wlds[NT * 16 * PITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lg = lane >> 4, li = lane & 15;

  // ---- weights -> LDS: row n (output channel), chunk q; rows >= Cout and chunks >= NCH are zeros ----
  {
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<uint16_t*>(g.W), 0, (int)g.w_bytes, 0x00020000);
    for (int i = tid; i < NT * 16 * KS * 4; i += 256) {
      const int n = i / (KS * 4), q = i - n * (KS * 4);
      const bool ok = n < g.Cout && q < NCH;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rW, ok ? (unsigned)(((int64_t)n * g.ldw + q * 8) * 2) : OOB, 0, 0);
      wlds[n * PITCH + q] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<uint16_t*>(g.in), 0, (int)g.in_bytes, 0x00020000);
  const bool out_f32 = g.flags & MIMO_EPI_OUT_F32;
  const bool do_silu = g.flags & MIMO_EPI_SILU;
  const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)g.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<float*>(g.bias), 0, g.bias ? g.Cout * 4 : 0, 0x00020000);
  f32x4 bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = nt * 16 + 4 * lg;
    bv[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, c < g.Cout ? (unsigned)c * 4u : OOB, 0, 0));
  }

  const int strips_x = (g.Wout + 15) >> 4;
  const int64_t nstrips = (int64_t)g.n * g.Hout * strips_x;
  for (int64_t sidx = (int64_t)blockIdx.x * 4 + wave; sidx < nstrips; sidx += (int64_t)gridDim.x * 4) {
    const int sx = (int)(sidx % strips_x);
    const int64_t t = sidx / strips_x;
    const int oy = (int)(t % g.Hout), img = (int)(t / g.Hout);
    const int ox = sx * 16 + li;
    const int iy0 = oy * g.stride - g.pad_t, ix0 = ox * g.stride - g.pad_l;
    // which of the 9 taps of this lane's pixel lie inside the image
    unsigned okmask = 0;
    if (ox < g.Wout) {
      if constexpr (TAPS == 1) {
        okmask = 1u;
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if ((unsigned)(iy0 + ky) < (unsigned)g.Hin && (unsigned)(ix0 + kx) < (unsigned)g.Win) okmask |= 1u << (ky * 3 + kx);
      }
    }
    // byte offset of tap (0, 0), channel 0 (modulo 2^32: exact for every in-range tap; the host keeps the image below 2 GiB)
    const unsigned base = (unsigned)((((int64_t)img * g.Hin + iy0) * g.Win + ix0) * CIN * 2);
    uint4 a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int q = 4 * s + lg;
      const int tap = q / CPT, cc = q - tap * CPT;
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;  // tap / 3 for tap < 12
      const bool ok = tap < TAPS && ((okmask >> tap) & 1u);  // (taps of the K tail: never set)
      const unsigned off = base + (unsigned)(((ky * g.Win + kx) * CIN + cc * 8) * 2);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rIn, ok ? off : OOB, 0, 0);
      a[s] = make_uint4(v.x, v.y, v.z, v.w);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = HT<DT>::mfma16(wlds[(nt * 16 + li) * PITCH + 4 * s + lg], a[s], acc[nt]);  // D[row = channel 4 lg + r][col = pixel li]
    if (ox < g.Wout) {
      const int64_t prow = (((int64_t)img * g.Hout + oy) * g.Wout + ox) * g.Cout;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = nt * 16 + 4 * lg;
        if (c >= g.Cout) continue;
        f32x4 v = acc[nt] + bv[nt];
        if (do_silu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
        }
        v *= g.out_scale;
        if (out_f32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rOut, (unsigned)((prow + c) * 4), 0, 0);
        } else {
          u32x2 o;
          o.x = pack2<DT>(v[0], v[1]); o.y = pack2<DT>(v[2], v[3]);
          __builtin_amdgcn_raw_buffer_store_b64(o, rOut, (unsigned)((prow + c) * 2), 0, 0);
        }
      }
    }
  }
}

template <int DT>
int launch_dt(const Args& a, unsigned grid, hipStream_t st) {
  const int nt = (a.Cout + 15) / 16;
#define THIN(CIN_, NT_, TAPS_) hipLaunchKernelGGL((thin_conv_kernel<DT, CIN_, NT_, TAPS_>), dim3(grid), dim3(256), 0, st, a)
  if (a.ksize == 1) {  // thin-N GEMM: N <= 48 (the tap GEMM of a thin-output convolution: 9 x 4 columns)
    if (a.Cin == 128) THIN(128, 3, 1); else THIN(320, 3, 1);
  } else if (a.Cin == 8) {
    if (nt == 1) THIN(8, 1, 9); else if (nt == 2) THIN(8, 2, 9); else THIN(8, 8, 9);
  } else if (a.Cin == 16) {
    if (nt == 1) THIN(16, 1, 9); else if (nt == 2) THIN(16, 2, 9); else THIN(16, 8, 9);
  } else {
    if (nt == 1) THIN(32, 1, 9); else THIN(32, 2, 9);
  }
#undef THIN
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

}  // namespace

bool supported(int Cin, int Cout, int ksize, int stride, int64_t in_bytes, int64_t out_bytes) {
  if (Cout <= 0 || (Cout & 3) || in_bytes >= 0x80000000LL || out_bytes >= 0x80000000LL) return false;
  if (ksize == 1) return stride == 1 && (Cin == 128 || Cin == 320) && Cout <= 48;
  if (ksize != 3) return false;
  if (Cin == 8 || Cin == 16) return Cout <= 128;
  return Cin == 32 && Cout <= 32;  // (32 -> 96: six of eight channel tiles used, the implicit-GEMM kernel is as fast)
}

int launch(int dtype, const Args& a, int cus, hipStream_t st) {
  const int64_t nstrips = (int64_t)a.n * a.Hout * ((a.Wout + 15) / 16);
  int64_t blocks = (nstrips + 3) / 4;
  const int64_t cap = (int64_t)cus * 8;  // 8 blocks of 4 waves per CU: each wave walks its strips
  if (blocks > cap) blocks = cap;
  if (blocks <= 0) return MIMO_EINVAL;
  if (dtype == MIMO_F16) return launch_dt<MIMO_F16>(a, (unsigned)blocks, st);
  if (dtype == MIMO_BF16) return launch_dt<MIMO_BF16>(a, (unsigned)blocks, st);
  return MIMO_EDTYPE;
}

}  // namespace mimo_thin
