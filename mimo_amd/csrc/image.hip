// image.hip — byte-exact image kernels either side of the denoising path (gfx950).  HBM-bound integer / byte work:
// one thread per output pixel, coalesced 3-byte pixels, no MFMA.
//
//   * resample_pass_kernel: one separable pass of Pillow's 8-bit resampler (ImagingResample, src/libImaging/Resample.c
//     of Pillow: fixed-point coefficients with PRECISION_BITS = 22, rounding constant 1 << 21, clip to [0, 255]).
//     The coefficient tables are produced on the host by mimo_amd/image.py exactly as Pillow's precompute_coeffs +
//     normalize_coeffs_8bpc do, so a horizontal + a vertical pass reproduce PIL.Image.resize bit for bit.
//     Replaces: VaeImageProcessor's PIL LANCZOS resize (pipeline_pose2vid_long_edit_bkfill_roiclip.py:424-457),
//     `ref_image.resize((224, 224))` + CLIPImageProcessor's bicubic resize (:379-384), and the
//     `res_image_pil.resize((pad_w, pad_h))` of the compositing loop (run_edit.py:268-269).  The source may be the
//     pipeline's fp32 video tensor: it is quantised on the fly as `(image * 255).astype(np.uint8)` (run_edit.py:267).
//   * u8_to_tokens_kernel: uint8 HWC image -> half16 channels-last tokens, x / 255 (fp32) and optionally 2 x - 1
//     (VaeImageProcessor.preprocess: numpy_to_pt(np.array(img).astype(float32) / 255), normalize), or
//     (x * rescale - mean) / std planar fp32 (CLIPImageProcessor rescale + normalize).
//   * composite_kernel: the per-frame compositing of run_edit.py:253-304 fused into one pass over the full frame:
//     un-pad crop -> paste on a white canvas -> alpha blend with the inpainted background (float32 arithmetic, as numpy) ->
//     occluder re-imposition (float64, as numpy: `occ / 255.0`) -> overlap cross-fade with the previous clip's frame
//     (float64) -> truncation to uint8.  Every arithmetic step is a separately rounded IEEE operation in numpy's order and
//     width (no FMA contraction), so the uint8 result equals the reference's.
#include "common.hip.h"

// numpy rounds every multiply and add separately: no FMA contraction anywhere in this file.  The arithmetic below is
// written with plain operators ON PURPOSE: the pragma governs expressions in this file, whereas the __fmul_rn / __dadd_rn
// helpers are inline functions compiled under the headers' own (contracting) mode and fuse again once inlined.
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow, Resample.c

struct ResampleArgs {
  const void* src;       // uint8 or fp32
  int src_f32;           // 1: fp32 source, quantised as (uint8)(v * 255.0f)
  int64_t sn, sy, sx, sc;  // source strides in elements: image, row, column, channel
  uint8_t* dst;          // uint8 [n, Hd, Wd, C] contiguous
  const int* bounds;     // [out_size][2]: first source index, tap count
  const int* kk;         // [out_size][ksize] fixed-point coefficients
  int ksize, horizontal, n, Hd, Wd, C;
};

__global__ __launch_bounds__(256) void resample_pass_kernel(const ResampleArgs a) {
  const int64_t total = (int64_t)a.n * a.Hd * a.Wd;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % a.Wd);
    const int64_t t = i / a.Wd;
    const int y = (int)(t % a.Hd);
    const int img = (int)(t / a.Hd);
    const int o = a.horizontal ? x : y;  // the output coordinate this pass resamples
    const int first = a.bounds[2 * o], count = a.bounds[2 * o + 1];
    const int* k = a.kk + (int64_t)o * a.ksize;
    const int64_t base = img * a.sn + (a.horizontal ? (int64_t)y * a.sy + (int64_t)first * a.sx
                                                    : (int64_t)first * a.sy + (int64_t)x * a.sx);
    const int64_t step = a.horizontal ? a.sx : a.sy;
    for (int c = 0; c < a.C; ++c) {
      int ss = 1 << (PRECISION_BITS - 1);
      for (int j = 0; j < count; ++j) {
        const int64_t idx = base + j * step + c * a.sc;
        int p;
        if (a.src_f32) p = (int)(uint8_t)(int)(((const float*)a.src)[idx] * 255.0f);
        else p = ((const uint8_t*)a.src)[idx];
        ss += p * k[j];
      }
      int v = ss >> PRECISION_BITS;  // arithmetic shift, then Pillow's clip8 lookup
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      a.dst[i * a.C + c] = (uint8_t)v;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void u8_to_tokens_kernel(const uint8_t* src, int64_t npix, int C, int cpad, int two_x_minus_1,
                                                           uint16_t* dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    for (int c = 0; c < cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        v = (float)src[i * C + c] / 255.0f;
        if (two_x_minus_1) v = 2.0f * v - 1.0f;
      }
      dst[i * cpad + c] = HT<DT>::from_f(v);
    }
  }
}

// the same values kept in fp32 (the VAE encoder's split-operand policy takes an fp32 image: no rounding of the input at all)
__global__ __launch_bounds__(256) void u8_to_tokens_f32_kernel(const uint8_t* src, int64_t npix, int C, int cpad, int two_x_minus_1,
                                                               float* dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    for (int c = 0; c < cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        v = (float)src[i * C + c] / 255.0f;
        if (two_x_minus_1) v = 2.0f * v - 1.0f;
      }
      dst[i * cpad + c] = v;
    }
  }
}

// uint8 HWC -> fp32 planar [n, C, H, W] = (x * rescale - mean[c]) / std[c]
__global__ __launch_bounds__(256) void u8_to_planar_kernel(const uint8_t* src, int n, int64_t HW, int C, float rescale,
                                                           const float* mean, const float* stdv, float* dst) {
  const int64_t total = (int64_t)n * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t img = i / HW, p = i - img * HW;
    for (int c = 0; c < C; ++c)
      dst[(img * C + c) * HW + p] = ((float)src[i * C + c] * rescale - mean[c]) / stdv[c];
  }
}

struct CompositeArgs {
  const uint8_t* crop;   // resized generated frame, uint8 [pad_h, pad_w, 3]
  int pad_h, pad_w, top, bottom, left, right;  // padding_v: the un-pad crop is [top, pad_h - bottom) x [left, pad_w - right)
  int w_min, h_min;      // paste position of the un-padded crop on the canvas
  const float* mask;     // float32 [mh, mw], placed at (h_min, w_min) of a zero mask_full
  int mh, mw;
  const uint8_t* bk;     // inpainted background frame, uint8 [H, W, 3]
  const uint8_t* occ;    // occluder mask frame uint8 [H, W, 3] (channel 0 is used) or null
  const uint8_t* vid;    // original video frame uint8 [H, W, 3] (with occ)
  const uint8_t* prev;   // previously composited frame (overlap cross-fade) or null
  double factor;         // (i - start_i + 1) / (overlay + 1)
  uint8_t* out;          // uint8 [H, W, 3]; may alias prev
  int H, W;
};

__global__ __launch_bounds__(256) void composite_kernel(const CompositeArgs a) {
  const int64_t total = (int64_t)a.H * a.W;
  const int ch = a.pad_h - a.top - a.bottom, cw = a.pad_w - a.left - a.right;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
    const int cy = y - a.h_min, cx = x - a.w_min;
    const bool in_crop = (cy >= 0) & (cy < ch) & (cx >= 0) & (cx < cw);
    const bool in_mask = (cy >= 0) & (cy < a.mh) & (cx >= 0) & (cx < a.mw);
    const float m = in_mask ? a.mask[(int64_t)cy * a.mw + cx] : 0.f;
    const float one_m = 1.0f - m;
    double o = 0.0, one_o = 1.0;
    if (a.occ) {
      o = (double)a.occ[i * 3] / 255.0;
      one_o = 1.0 - o;
    }
    for (int c = 0; c < 3; ++c) {
      const float canvas = in_crop ? (float)a.crop[((int64_t)(cy + a.top) * a.pad_w + (cx + a.left)) * 3 + c] : 255.0f;
      // res_image * mask_full[..., None] + bk_image * (1 - mask_full[..., None]): float32
      const float r32 = canvas * m + (float)a.bk[i * 3 + c] * one_m;
      double r;
      bool r_is_f64 = false;
      if (a.occ) {  // res_image * (1 - occ) + vid_image * occ: float64
        r = (double)r32 * one_o + (double)a.vid[i * 3 + c] * o;
        r_is_f64 = true;
      } else {
        r = (double)r32;
      }
      uint8_t q;
      if (a.prev) {
        // res_images[i] * (1 - factor) + res_image * factor: the uint8 array times a Python float is float64; a float32
        // res_image times a Python float stays float32 (the scalar is cast), a float64 one float64
        const double pterm = (double)a.prev[i * 3 + c] * (1.0 - a.factor);
        const double rterm = r_is_f64 ? r * a.factor : (double)(r32 * (float)a.factor);
        q = (uint8_t)(int)(pterm + rterm);
      } else {
        q = r_is_f64 ? (uint8_t)(int)r : (uint8_t)(int)r32;
      }
      a.out[i * 3 + c] = q;
    }
  }
}

inline unsigned grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int mimo_resample_pass_u8(const void* src, int src_is_f32, int64_t src_stride_n, int64_t src_stride_y,
                                     int64_t src_stride_x, int64_t src_stride_c, void* dst, int n, int Hd, int Wd, int C,
                                     const int* bounds, const int* coeffs, int ksize, int horizontal, void* stream) {
  if (!src || !dst || !bounds || !coeffs || n <= 0 || Hd <= 0 || Wd <= 0 || C <= 0 || C > 4 || ksize <= 0) return MIMO_EINVAL;
  ResampleArgs a;
  a.src = src; a.src_f32 = src_is_f32; a.sn = src_stride_n; a.sy = src_stride_y; a.sx = src_stride_x; a.sc = src_stride_c;
  a.dst = (uint8_t*)dst; a.bounds = bounds; a.kk = coeffs; a.ksize = ksize; a.horizontal = horizontal;
  a.n = n; a.Hd = Hd; a.Wd = Wd; a.C = C;
  hipLaunchKernelGGL(resample_pass_kernel, dim3(grid_for((int64_t)n * Hd * Wd)), dim3(256), 0, (hipStream_t)stream, a);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_u8_to_tokens(int dtype, const void* src, int64_t npix, int C, int Cpad, int two_x_minus_1, void* dst,
                                 void* stream) {
  if (!src || !dst || npix <= 0 || C <= 0 || Cpad < C) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(u8_to_tokens_kernel<MIMO_F16>, dim3(grid_for(npix)), dim3(256), 0, st, (const uint8_t*)src, npix, C, Cpad, two_x_minus_1, (uint16_t*)dst);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(u8_to_tokens_kernel<MIMO_BF16>, dim3(grid_for(npix)), dim3(256), 0, st, (const uint8_t*)src, npix, C, Cpad, two_x_minus_1, (uint16_t*)dst);
  else if (dtype == MIMO_F32)
    hipLaunchKernelGGL(u8_to_tokens_f32_kernel, dim3(grid_for(npix)), dim3(256), 0, st, (const uint8_t*)src, npix, C, Cpad, two_x_minus_1, (float*)dst);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_u8_to_planar_f32(const void* src, int n, int64_t HW, int C, float rescale, const float* mean,
                                     const float* stdv, float* dst, void* stream) {
  if (!src || !dst || !mean || !stdv || n <= 0 || HW <= 0 || C <= 0) return MIMO_EINVAL;
  hipLaunchKernelGGL(u8_to_planar_kernel, dim3(grid_for((int64_t)n * HW)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)src, n, HW, C, rescale, mean, stdv, dst);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_composite_frame(const mimo_composite_params* p, void* stream) {
  if (!p || !p->crop || !p->mask || !p->bk || !p->out || p->H <= 0 || p->W <= 0) return MIMO_EINVAL;
  if (p->pad_h <= 0 || p->pad_w <= 0 || p->top < 0 || p->bottom < 0 || p->left < 0 || p->right < 0) return MIMO_EINVAL;
  if (p->top + p->bottom >= p->pad_h || p->left + p->right >= p->pad_w) return MIMO_EINVAL;
  if (p->occ && !p->vid) return MIMO_EINVAL;
  if (p->mh < 0 || p->mw < 0) return MIMO_EINVAL;
  CompositeArgs a;
  a.crop = (const uint8_t*)p->crop; a.pad_h = p->pad_h; a.pad_w = p->pad_w; a.top = p->top; a.bottom = p->bottom;
  a.left = p->left; a.right = p->right; a.w_min = p->w_min; a.h_min = p->h_min; a.mask = p->mask; a.mh = p->mh; a.mw = p->mw;
  a.bk = (const uint8_t*)p->bk; a.occ = (const uint8_t*)p->occ; a.vid = (const uint8_t*)p->vid;
  a.prev = (const uint8_t*)p->prev; a.factor = p->factor; a.out = (uint8_t*)p->out; a.H = p->H; a.W = p->W;
  hipLaunchKernelGGL(composite_kernel, dim3(grid_for((int64_t)p->H * p->W)), dim3(256), 0, (hipStream_t)stream, a);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}
