// elementwise.hip — boundary layout conversions and the fused per-step latent update.
// Reference ops replaced: see include/mimo_hip.h.  All HBM-bound, tiny tensors
// (latents are 0.4 MB per 24-frame window); one thread per element, grid-stride.
#include "common.hip.h"

namespace {

inline unsigned ew_grid(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <int DT>
__global__ __launch_bounds__(256) void ncfhw_to_tokens_kernel(const void* in, int f32, int b, int C, int F,
                                                              int64_t HW, const int* frame_idx, int Fsel,
                                                              int Cpad, int64_t out_ld, int out_col0,
                                                              uint16_t* out) {
  const int64_t total = (int64_t)b * Fsel * HW * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const int64_t tok = i / Cpad;
    const int64_t p = tok % HW;
    const int64_t bj = tok / HW;
    const int j = (int)(bj % Fsel);
    const int bi = (int)(bj / Fsel);
    float v = 0.f;
    if (c < C) {
      const int f = frame_idx ? frame_idx[j] : j;
      const int64_t src = (((int64_t)bi * C + c) * F + f) * HW + p;
      v = f32 ? ((const float*)in)[src] : HT<DT>::to_f(((const uint16_t*)in)[src]);
    }
    out[tok * out_ld + out_col0 + c] = HT<DT>::from_f(v);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void tokens_to_ncfhw_kernel(const void* in, int f32, int64_t ld, int b, int C,
                                                              int F, int64_t HW, float scale, float* out) {
  const int64_t total = (int64_t)b * C * F * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    int64_t r = i / HW;
    const int f = (int)(r % F);
    r /= F;
    const int c = (int)(r % C);
    const int bi = (int)(r / C);
    const int64_t src = (((int64_t)bi * F + f) * HW + p) * ld + c;
    const float v = f32 ? ((const float*)in)[src] : HT<DT>::to_f(((const uint16_t*)in)[src]);
    out[i] = v * scale;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void tokens_to_image_kernel(const void* in, int f32, int64_t ld, int n,
                                                              int64_t HW, float* out) {
  const int64_t total = (int64_t)n * 3 * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    const int64_t r = i / HW;
    const int c = (int)(r % 3);
    const int64_t im = r / 3;
    const int64_t src = (im * HW + p) * ld + c;
    const float v = f32 ? ((const float*)in)[src] : HT<DT>::to_f(((const uint16_t*)in)[src]);
    out[i] = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);
  }
}

__global__ __launch_bounds__(256) void window_accumulate_kernel(const float* pred, int64_t ld, const int* frames,
                                                                int Fw, int bb, int C, int F, int64_t HW,
                                                                float* acc, float* counter) {
  const int64_t total = (int64_t)bb * C * Fw * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i % HW;
    int64_t r = i / HW;
    const int j = (int)(r % Fw);
    r /= Fw;
    const int c = (int)(r % C);
    const int bi = (int)(r / C);
    const int f = frames[j];
    if (f < 0) continue;  // frame not taken from this prediction (a frame segment of the window only, see the header)
    acc[(((int64_t)bi * C + c) * F + f) * HW + p] += pred[(((int64_t)bi * Fw + j) * HW + p) * ld + c];
  }
  if (blockIdx.x == 0 && threadIdx.x < Fw && frames[threadIdx.x] >= 0) counter[frames[threadIdx.x]] += 1.f;
}

__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* acc, const float* counter, float* lat, int C,
                                                       int F, int64_t HW, int cfg, float guidance, float sa,
                                                       float s1, float sap, float s1p) {
  const int64_t total = (int64_t)C * F * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float np;
    if (cfg) {
      const int f = (int)((i / HW) % F);
      const float cnt = counter[f];
      const float un = acc[i] / cnt;
      const float co = acc[total + i] / cnt;
      np = un + guidance * (co - un);
    } else {
      // reference quirk: without CFG the window average is skipped
      // (pipeline_pose2vid_long_edit_bkfill_roiclip.py:545-549)
      np = acc[i];
    }
    const float x = lat[i];
    const float x0 = sa * x - s1 * np;
    const float eps = sa * np + s1 * x;
    lat[i] = sap * x0 + s1p * eps;
  }
}

// The same update restricted to a list of frames (the cross-step schedule of the sharded long clip advances a frame as soon as
// every window that covers it has delivered its prediction; per element the arithmetic is cfg_ddim_kernel's, bit for bit).
__global__ __launch_bounds__(256) void cfg_ddim_frames_kernel(const float* acc, const float* counter, float* lat, int C,
                                                              int F, int64_t HW, const int* frames, int nf, int cfg,
                                                              float guidance, float sa, float s1, float sap, float s1p) {
  const int64_t total = (int64_t)C * nf * HW;
  const int64_t plane = (int64_t)C * F * HW;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
    const int64_t p = j % HW;
    const int64_t r = j / HW;
    const int f = frames[(int)(r % nf)];
    const int c = (int)(r / nf);
    const int64_t i = ((int64_t)c * F + f) * HW + p;
    float np;
    if (cfg) {
      const float cnt = counter[f];
      const float un = acc[i] / cnt;
      const float co = acc[plane + i] / cnt;
      np = un + guidance * (co - un);
    } else {
      np = acc[i];
    }
    const float x = lat[i];
    const float x0 = sa * x - s1 * np;
    const float eps = sa * np + s1 * x;
    lat[i] = sap * x0 + s1p * eps;
  }
}

// differ[i] = 1 where frame i is not bit-identical to frame i - 1 (differ[0] = 1): 16-byte compares, one flag store per
// block that finds a difference (benign race: every writer stores the same value).  The caller zero-fills differ[1..].
__global__ __launch_bounds__(256) void frames_differ_kernel(const uint4* x, int64_t vec_per_frame, int blocks_per_frame, int* differ) {
  const int f = 1 + blockIdx.x / blocks_per_frame, sl = blockIdx.x % blocks_per_frame;
  if (blockIdx.x == 0 && threadIdx.x == 0) differ[0] = 1;
  const uint4* a = x + (int64_t)f * vec_per_frame;
  const uint4* b = a - vec_per_frame;
  bool d = false;
  for (int64_t i = (int64_t)sl * 256 + threadIdx.x; i < vec_per_frame; i += (int64_t)blocks_per_frame * 256) {
    const uint4 u = a[i], v = b[i];
    d |= (u.x != v.x) | (u.y != v.y) | (u.z != v.z) | (u.w != v.w);
  }
  if (__any(d) && (threadIdx.x & 63) == 0) differ[f] = 1;
}

// Second half of a thin-output 3x3 convolution (Cout <= 16: conv_out of the UNet and of the VAE decoder).  The first half is a
// plain GEMM of the input pixels with the weight regrouped as [9 taps x Cout][Cin] (mimo_amd.packing.pack_conv_taps): every
// pixel's contribution to each of the nine output positions it touches, taps[m][tap * Cout + c], the input read ONCE.  This
// kernel gathers them: out[y][x][c] = bias[c] + sum_tap taps[(y + ky - 1, x + kx - 1)][tap][c] (fixed tap order, zero outside
// the image).  A block owns TH x 32 output pixels (TH = 8 / (Cout / 4)): the (TH + 2) x 34 tap rows they need come in as whole
// contiguous rows (lane = consecutive 16-byte piece: full cache lines; a per-pixel gather would pull a 128-byte line for every
// 16 bytes it uses) into LDS, then one thread per (pixel, 4 channels) sums its nine pieces.
template <int C4>
__global__ __launch_bounds__(256) void conv3x3_tapsum_kernel(const float* taps, int64_t ldt, int n, int H, int W,
                                                             const float* bias, float* out, float out_scale) {
  constexpr int TH = 8 / C4, TW = 32, COUT = 4 * C4;
  constexpr int PCS = 9 * C4;                     // 16-byte pieces per tap row
  constexpr int NPIX = (TH + 2) * (TW + 2);
  __shared__ float4 tile[NPIX * PCS];
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int bx = blockIdx.x % tiles_x;
  const int by = (blockIdx.x / tiles_x) % tiles_y;
  const int64_t img = blockIdx.x / (tiles_x * tiles_y);
  const int y0 = by * TH, x0 = bx * TW;
  for (int i = threadIdx.x; i < NPIX * PCS; i += 256) {
    const int pix = i / PCS, pc = i - pix * PCS;
    const int r = pix / (TW + 2), c = pix - r * (TW + 2);
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
      v = *reinterpret_cast<const float4*>(taps + ((img * H + yy) * W + xx) * ldt + 4 * pc);
    tile[i] = v;
  }
  __syncthreads();
  const int cq = threadIdx.x % C4, tx = (threadIdx.x / C4) % TW, ty = threadIdx.x / (C4 * TW);
  const int y = y0 + ty, x = x0 + tx;
  if (y >= H || x >= W) return;
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float4 v = tile[((ty + ky) * (TW + 2) + tx + kx) * PCS + (ky * 3 + kx) * C4 + cq];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  acc.x *= out_scale; acc.y *= out_scale; acc.z *= out_scale; acc.w *= out_scale;
  *reinterpret_cast<float4*>(out + ((img * H + y) * W + x) * COUT + 4 * cq) = acc;
}

}  // namespace

extern "C" int mimo_ncfhw_to_tokens(const void* in, int in_is_f32, int dtype, int b, int C, int F, int H,
                                    int W, const int* frame_idx, int Fsel, int Cpad, int64_t out_ld,
                                    int out_col0, void* out, void* stream) {
  if (!in || !out || b <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || Fsel <= 0 || Cpad < C) return MIMO_EINVAL;
  if (out_col0 < 0 || out_col0 + Cpad > out_ld) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t HW = (int64_t)H * W;
  const unsigned grid = ew_grid((int64_t)b * Fsel * HW * Cpad);
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(ncfhw_to_tokens_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, b, C, F, HW, frame_idx, Fsel, Cpad, out_ld, out_col0, (uint16_t*)out);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(ncfhw_to_tokens_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, b, C, F, HW, frame_idx, Fsel, Cpad, out_ld, out_col0, (uint16_t*)out);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_tokens_to_ncfhw(const void* in, int in_is_f32, int dtype, int64_t ld, int b, int C, int F,
                                    int H, int W, float scale, float* out, void* stream) {
  if (!in || !out || b <= 0 || C <= 0 || F <= 0 || H <= 0 || W <= 0 || ld < C) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t HW = (int64_t)H * W;
  const unsigned grid = ew_grid((int64_t)b * C * F * HW);
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(tokens_to_ncfhw_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, ld, b, C, F, HW, scale, out);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(tokens_to_ncfhw_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, ld, b, C, F, HW, scale, out);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_tokens_to_image(const void* in, int in_is_f32, int dtype, int64_t ld, int n, int H, int W,
                                    float* out, void* stream) {
  if (!in || !out || n <= 0 || H <= 0 || W <= 0 || ld < 3) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t HW = (int64_t)H * W;
  const unsigned grid = ew_grid((int64_t)n * 3 * HW);
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(tokens_to_image_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, ld, n, HW, out);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(tokens_to_image_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, ld, n, HW, out);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_window_accumulate(const float* pred, int64_t ld, const int* frames, int Fw, int bb, int C,
                                      int F, int64_t HW, float* acc, float* counter, void* stream) {
  if (!pred || !frames || !acc || !counter || Fw <= 0 || Fw > 256 || bb <= 0 || C <= 0 || F <= 0 || HW <= 0 || ld < C)
    return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(window_accumulate_kernel, dim3(ew_grid((int64_t)bb * C * Fw * HW)), dim3(256), 0, st, pred, ld, frames, Fw, bb, C, F, HW, acc, counter);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_cfg_ddim_step(const float* acc, const float* counter, float* latents, int C, int F,
                                  int64_t HW, int cfg, float guidance, float sqrt_a_t, float sqrt_1ma_t,
                                  float sqrt_a_prev, float sqrt_1ma_prev, void* stream) {
  if (!acc || !counter || !latents || C <= 0 || F <= 0 || HW <= 0) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(ew_grid((int64_t)C * F * HW)), dim3(256), 0, st, acc, counter, latents, C, F, HW, cfg, guidance, sqrt_a_t, sqrt_1ma_t, sqrt_a_prev, sqrt_1ma_prev);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_cfg_ddim_step_frames(const float* acc, const float* counter, float* latents, int C, int F, int64_t HW,
                                         const int* frames, int nf, int cfg, float guidance, float sqrt_a_t, float sqrt_1ma_t,
                                         float sqrt_a_prev, float sqrt_1ma_prev, void* stream) {
  if (!acc || !counter || !latents || !frames || C <= 0 || F <= 0 || HW <= 0 || nf <= 0 || nf > F) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cfg_ddim_frames_kernel, dim3(ew_grid((int64_t)C * nf * HW)), dim3(256), 0, st, acc, counter, latents, C, F, HW,
                     frames, nf, cfg, guidance, sqrt_a_t, sqrt_1ma_t, sqrt_a_prev, sqrt_1ma_prev);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_frames_differ(const void* frames, int n, int64_t frame_bytes, int* differ, void* stream) {
  if (!frames || !differ || n <= 0 || frame_bytes <= 0 || (frame_bytes & 15) || (reinterpret_cast<uintptr_t>(frames) & 15u)) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (n == 1) {
    hipLaunchKernelGGL(frames_differ_kernel, dim3(1), dim3(64), 0, st, (const uint4*)frames, (int64_t)0, 1, differ);
    MIMO_LAUNCH_CHECK();
    return MIMO_OK;
  }
  const int64_t vec = frame_bytes / 16;
  int bpf = (int)((vec + 256 * 16 - 1) / (256 * 16));   // >= 16 vectors per thread
  if (bpf < 1) bpf = 1;
  if (bpf > 256) bpf = 256;
  hipLaunchKernelGGL(frames_differ_kernel, dim3((unsigned)((n - 1) * bpf)), dim3(256), 0, st, (const uint4*)frames, vec, bpf, differ);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_conv3x3_tapsum(const float* taps, int64_t ldt, int n, int H, int W, int cout, const float* bias, float* out,
                                   float out_scale, void* stream) {
  if (!taps || !out || n <= 0 || H <= 0 || W <= 0 || cout <= 0 || (cout & 3) || cout > 16 || ldt < 9 * (int64_t)cout || (ldt & 3))
    return MIMO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(taps) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)))
    return MIMO_EINVAL;
  const int c4 = cout >> 2;
  if (c4 != 1 && c4 != 2 && c4 != 4) return MIMO_EINVAL;
  const int th = 8 / c4;
  const int64_t blocks = (int64_t)n * ((H + th - 1) / th) * ((W + 31) / 32);
  if (blocks > 0x7fffffff) return MIMO_EINVAL;
  const dim3 gr((unsigned)blocks), bl(256);
  hipStream_t st = (hipStream_t)stream;
  if (c4 == 1) hipLaunchKernelGGL(conv3x3_tapsum_kernel<1>, gr, bl, 0, st, taps, ldt, n, H, W, bias, out, out_scale);
  else if (c4 == 2) hipLaunchKernelGGL(conv3x3_tapsum_kernel<2>, gr, bl, 0, st, taps, ldt, n, H, W, bias, out, out_scale);
  else hipLaunchKernelGGL(conv3x3_tapsum_kernel<4>, gr, bl, 0, st, taps, ldt, n, H, W, bias, out, out_scale);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}
