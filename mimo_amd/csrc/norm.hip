// norm.hip — HBM-bound normalisation kernels of the denoising path (gfx950).
//   GroupNorm statistics + apply(+SiLU) over a virtual channel concat of two sources,
//   LayerNorm (+ temporal positional-encoding add), casts.
// Reference ops replaced: see include/mimo_hip.h.
// All are streaming kernels: 16-byte vector loads, fp32 statistics, no MFMA.  GroupNorm statistics
// are one-pass SHIFTED sums (shift = the group's first element) or, when the producing GEMM / convolution
// emitted column statistics in its epilogue, a merge of those (gn_stats_cols_kernel: no pass over the
// tensor at all); LayerNorm is two-pass (mean, then centred variance) on a row held in registers.
#include "common.hip.h"

namespace {

template <int DT>
__device__ __forceinline__ float load_elem(const void* p, bool f32, int64_t idx) {
  return f32 ? ((const float*)p)[idx] : HT<DT>::to_f(((const uint16_t*)p)[idx]);
}

// ---------------------------------------------------------------------------------
// GroupNorm statistics: one block per (image, group), ONE pass over the group's elements with
// shifted sums (shift = the group's first element, which removes the E[x^2]-E[x]^2 cancellation),
// 2-element vector loads, division-free incremental indexing.
// ---------------------------------------------------------------------------------
template <int DT, int V>  // V = elements per load (2 when channels-per-group and C1 are even, else 1)
__global__ __launch_bounds__(256) void gn_stats_kernel(const void* x1, int C1, const void* x2, int C2,
                                                       int f32, int64_t HW, int groups, float eps,
                                                       float* stats, float* partials, int split) {
  // block -> (image, group, pixel slice): `split` blocks share one (image, group) when n*groups alone cannot
  // fill the chip (VAE: 8 images x 32 groups over 1 GB); their (S, Q) partials are reduced in fixed order by
  // gn_finalize_kernel, so the result does not depend on scheduling
  const int ig = blockIdx.x / split, sl = blockIdx.x - ig * split;
  const int img = ig / groups;
  const int grp = ig % groups;
  const int C = C1 + C2;
  const int cpg = C / groups;
  const int c0 = grp * cpg;
  const int hp = cpg / V;              // loads per pixel
  const int count2 = (int)(HW * hp);   // host guarantees < 2^31
  __shared__ float red[8];

  auto fetch2 = [&](int p, int c, float& a, float& b) {
    const bool first = c < C1;
    const void* src = first ? x1 : x2;
    const int64_t idx = first ? ((int64_t)img * HW + p) * C1 + c : ((int64_t)img * HW + p) * C2 + (c - C1);
    if (V == 2) {
      if (f32) {
        const float2 v = *reinterpret_cast<const float2*>((const float*)src + idx);
        a = v.x; b = v.y;
      } else {
        const uint32_t v = *reinterpret_cast<const uint32_t*>((const uint16_t*)src + idx);
        a = HT<DT>::to_f((uint16_t)(v & 0xffffu)); b = HT<DT>::to_f((uint16_t)(v >> 16));
      }
    } else {
      a = f32 ? ((const float*)src)[idx] : HT<DT>::to_f(((const uint16_t*)src)[idx]);
      b = a;
    }
  };
  float sh, sh2;
  fetch2(0, c0, sh, sh2);  // same value in every thread
  const int dp = 256 / hp, dc = 256 - dp * hp;
  const int per = ((int)HW + split - 1) / split;              // pixels per slice
  const int p_lo = sl * per, p_hi = min((int)HW, p_lo + per);
  const int e_hi = p_hi * hp;
  int p = p_lo + threadIdx.x / hp, c2 = threadIdx.x % hp;
  float s = 0.f, q = 0.f;
  for (int e = p_lo * hp + threadIdx.x; e < e_hi; e += 256) {
    float a, b;
    fetch2(p, c0 + V * c2, a, b);
    a -= sh; b -= sh;
    if (V == 2) {
      s += a + b;
      q += a * a + b * b;
    } else {
      s += a;
      q += a * a;
    }
    p += dp; c2 += dc;
    if (c2 >= hp) { c2 -= hp; ++p; }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S = red[0] + red[1] + red[2] + red[3], Q = red[4] + red[5] + red[6] + red[7];
    if (split > 1) {
      partials[(int64_t)blockIdx.x * 2 + 0] = S;  // sums of (x - sh), (x - sh)^2 over this slice; sh is slice-independent
      partials[(int64_t)blockIdx.x * 2 + 1] = Q;
    } else {
      const float n = (float)count2 * (float)V;
      const float ms = S / n;
      const float var = fmaxf(Q / n - ms * ms, 0.f);
      stats[(int64_t)ig * 2 + 0] = sh + ms;
      stats[(int64_t)ig * 2 + 1] = rsqrtf(var + eps);
    }
  }
}

// GroupNorm statistics, row-streaming form: one block per (image, pixel slice) covers ALL groups.  A thread owns
// 4 consecutive channels and walks the slice's pixels, so every wave instruction reads whole contiguous pixel rows
// (the per-(image, group) form above touches 4*cpg bytes out of every C*4-byte row and makes each XCD's L2 fetch
// every line).  Same shifted sums (shift = the group's first element of the image), reduced per group in a fixed
// order and written as (S, Q) partials for gn_finalize_kernel.
template <int DT>
__global__ __launch_bounds__(1024) void gn_stats_rows_kernel(const void* x1, int C1, const void* x2, int C2, int f32,
                                                             int64_t HW, int groups, float* partials, int split) {
  const int C = C1 + C2, cpg = C / groups, c4 = C >> 2;
  const int R = blockDim.x / c4;  // pixels per block pass
  const int img = blockIdx.x / split, sl = blockIdx.x - img * split;
  const int per = ((int)HW + split - 1) / split;
  const int p_lo = sl * per, p_hi = min((int)HW, p_lo + per);
  const int tid = threadIdx.x;
  const int r = tid / c4, ch = (tid - r * c4) * 4;
  const bool first = ch < C1;
  const int Cx = first ? C1 : C2, cx = first ? ch : ch - C1;
  const void* src = first ? x1 : x2;
  const int64_t img_base = (int64_t)img * HW * Cx + cx;
  float sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int g0 = ((ch + j) / cpg) * cpg;  // first channel of this channel's group
    sh[j] = g0 < C1 ? load_elem<DT>(x1, f32, (int64_t)img * HW * C1 + g0)
                    : load_elem<DT>(x2, f32, (int64_t)img * HW * C2 + (g0 - C1));
  }
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int p, float (&v)[4]) {
    const int64_t idx = img_base + (int64_t)p * Cx;
    if (f32) {
      const float4 t = *reinterpret_cast<const float4*>((const float*)src + idx);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      const uint2 t = *reinterpret_cast<const uint2*>((const uint16_t*)src + idx);
      v[0] = HT<DT>::to_f((uint16_t)(t.x & 0xffffu)); v[1] = HT<DT>::to_f((uint16_t)(t.x >> 16));
      v[2] = HT<DT>::to_f((uint16_t)(t.y & 0xffffu)); v[3] = HT<DT>::to_f((uint16_t)(t.y >> 16));
    }
  };
  auto accum = [&](const float (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = v[j] - sh[j];
      s[j] += a;
      q[j] = fmaf(a, a, q[j]);
    }
  };
  int p = p_lo + r;
  for (; p + 3 * R < p_hi; p += 4 * R) {  // four rows in flight per thread
    float v0[4], v1[4], v2[4], v3[4];
    fetch(p, v0); fetch(p + R, v1); fetch(p + 2 * R, v2); fetch(p + 3 * R, v3);
    accum(v0); accum(v1); accum(v2); accum(v3);
  }
  for (; p < p_hi; p += R) {
    float v0[4];
    fetch(p, v0);
    accum(v0);
  }
  // per-group reduction in a fixed order: channel-major, then pixel-row r
  extern __shared__ float red_rows[];  // [blockDim.x][8]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red_rows[tid * 8 + j] = s[j];
    red_rows[tid * 8 + 4 + j] = q[j];
  }
  __syncthreads();
  if (tid < groups) {
    float S = 0.f, Q = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
      for (int rr = 0; rr < R; ++rr) {
        const int t = rr * c4 + (c >> 2);
        S += red_rows[t * 8 + (c & 3)];
        Q += red_rows[t * 8 + 4 + (c & 3)];
      }
    const int64_t o = (((int64_t)img * groups + tid) * split + sl) * 2;
    partials[o] = S;
    partials[o + 1] = Q;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const void* x1, int C1, const void* x2, int C2, int f32,
                                                          int64_t HW, int groups, int total, float eps,
                                                          const float* partials, int split, float* stats) {
  // one wave per (image, group): lane l adds partials l, l + 64, ... in double, then a fixed butterfly — the result
  // depends on `split` only, never on scheduling
  const int ig = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (ig >= total) return;
  const int img = ig / groups, grp = ig % groups;
  const int cpg = (C1 + C2) / groups, c0 = grp * cpg;
  double S = 0.0, Q = 0.0;
  for (int k = lane; k < split; k += 64) {
    S += (double)partials[((int64_t)ig * split + k) * 2];
    Q += (double)partials[((int64_t)ig * split + k) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    S += __shfl_xor(S, o, 64);
    Q += __shfl_xor(Q, o, 64);
  }
  if (lane == 0) {
    // the shift every slice used: the group's first element
    const float sh = c0 < C1 ? load_elem<DT>(x1, f32, (int64_t)img * HW * C1 + c0)
                             : load_elem<DT>(x2, f32, (int64_t)img * HW * C2 + (c0 - C1));
    const double n = (double)HW * cpg;
    const double ms = S / n;
    const double var = fmax(Q / n - ms * ms, 0.0);
    stats[(int64_t)ig * 2 + 0] = sh + (float)ms;
    stats[(int64_t)ig * 2 + 1] = rsqrtf((float)var + eps);
  }
}


// ---------------------------------------------------------------------------------
// GroupNorm statistics from epilogue column statistics (mimo_epilogue_ext.colstats of the producing GEMM /
// convolution): cs[slab][0][c] = mean, cs[slab][1][c] = sum of squared deviations of the slab's rows of column c.
// One block per (image, group) merges (slabs of the image) x (columns of the group) in two passes over the partials
// (which sit in L2): the count-weighted mean of the slab means, then sum(m2_i + rows_i * (mean_i - mean)^2) — the
// pairwise update of Chan et al. summed in closed form, no division per item; double accumulators, a fixed reduction
// order (thread-sequential, xor butterfly inside a wave, waves in order): the result depends on the data only.
// The two sources of a virtual concat may come with different slab sizes (32-row slabs of the GEMM / conv epilogues,
// 256-pixel tiles of mimo_conv3x3_fused).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_cols_kernel(const float* cs1, int C1, int rows1, const float* cs2, int C2, int rows2,
                                                            int64_t HW, int groups, float eps, float* stats) {
  const int ig = blockIdx.x;
  const int img = ig / groups, grp = ig % groups;
  const int cpg = (C1 + C2) / groups, c0 = grp * cpg, c1 = c0 + cpg;
  __shared__ double red[4];
  auto block_sum = [&](double v) -> double {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();  // (red is reused by the second pass)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
  };
  // columns [lo, hi) of a source with Cx columns and `rows` pixels per slab; f(rows, mean_i, m2_i) summed over the items
  auto walk = [&](const float* cs, int Cx, int lo, int hi, int rows, auto f) -> double {
    const int nch = hi - lo;
    double acc = 0.0;
    if (nch <= 0) return acc;
    const int64_t slabs = HW / rows, items = slabs * nch;
    const float* base = cs + img * slabs * 2 * Cx + lo;
    for (int64_t i = threadIdx.x; i < items; i += 256) {
      const float* p = base + (i / nch) * 2 * Cx + (int)(i % nch);
      acc += f((double)rows, (double)p[0], (double)p[Cx]);
    }
    return acc;
  };
  const int lo1 = min(c0, C1), hi1 = min(c1, C1), lo2 = max(c0, C1) - C1, hi2 = max(c1, C1) - C1;
  const double n = (double)HW * (double)cpg;
  auto wsum = [](double r, double m, double) { return r * m; };
  const double mean = block_sum(walk(cs1, C1, lo1, hi1, rows1, wsum) + walk(cs2, C2, lo2, hi2, rows2, wsum)) / n;
  auto dev2 = [mean](double r, double m, double m2) { return m2 + r * (m - mean) * (m - mean); };
  const double q = block_sum(walk(cs1, C1, lo1, hi1, rows1, dev2) + walk(cs2, C2, lo2, hi2, rows2, dev2));
  if (threadIdx.x == 0) {
    stats[(int64_t)ig * 2 + 0] = (float)mean;
    stats[(int64_t)ig * 2 + 1] = rsqrtf((float)(q / n) + eps);
  }
}

// ---------------------------------------------------------------------------------
// GroupNorm apply: each thread handles 8 consecutive channels of one token.
// ---------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* x1, int C1, const void* x2, int C2,
                                                       int f32, int n, int64_t HW, int groups,
                                                       const float* stats, const float* gamma,
                                                       const float* beta, int silu, uint16_t* out,
                                                       uint16_t* raw) {
  const int C = C1 + C2;
  const int cpg = C / groups;
  const int c8 = C / 8;
  const int64_t total = (int64_t)n * HW * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t tok = i / c8;
    const int c = (int)(i - tok * c8) * 8;
    const int img = (int)(tok / HW);
    float v[8];
    // C1 % 8 == 0 is enforced by the host, so a chunk never straddles the two sources
    const void* src = c < C1 ? x1 : x2;
    const int64_t off = c < C1 ? tok * C1 + c : tok * C2 + (c - C1);
    if (f32) {
      const float4 a = *reinterpret_cast<const float4*>((const float*)src + off);
      const float4 b = *reinterpret_cast<const float4*>((const float*)src + off + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const uint4 a = *reinterpret_cast<const uint4*>((const uint16_t*)src + off);
      unpack8<DT>(a, v);
    }
    if (raw) *reinterpret_cast<uint4*>(raw + tok * C + c) = pack8<DT>(v);
    if (out) {
      float y[8];
      // an 8-channel chunk spans at most two groups (cpg >= 4 in every model on this path; general fallback below)
      const int g0 = c / cpg;
      const int split = (g0 + 1) * cpg - c;  // first index k that belongs to the next group
      const float2 s0 = *reinterpret_cast<const float2*>(stats + ((int64_t)img * groups + g0) * 2);
      const float2 s1 = *reinterpret_cast<const float2*>(stats + ((int64_t)img * groups + min(g0 + 1, groups - 1)) * 2);
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c), gb = *reinterpret_cast<const float4*>(gamma + c + 4);
      const float4 ba = *reinterpret_cast<const float4*>(beta + c), bb = *reinterpret_cast<const float4*>(beta + c + 4);
      const float gm[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
      const float bt[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float mean, rstd;
        if (cpg >= 8 || k < split + cpg) {
          mean = k < split ? s0.x : s1.x;
          rstd = k < split ? s0.y : s1.y;
        } else {  // tiny groups (cpg < 7): a chunk may span more than two
          const int grp = (c + k) / cpg;
          mean = stats[((int64_t)img * groups + grp) * 2];
          rstd = stats[((int64_t)img * groups + grp) * 2 + 1];
        }
        const float t = (v[k] - mean) * rstd * gm[k] + bt[k];
        y[k] = silu ? silu_f(t) : t;
      }
      *reinterpret_cast<uint4*>(out + tok * C + c) = pack8<DT>(y);
    }
  }
}

// ---------------------------------------------------------------------------------
// Split-operand form of the apply pass (the VAE's "split" precision policy): y = silu?(GroupNorm(x)) (or y = x without
// statistics) is written as THREE half16 channel blocks per token, [hi | hi | lo] with hi = half(y), lo = half(y - hi).
// Against a weight packed as [Whi | Wlo | Whi] along K (packing.pack_conv_split3) one ordinary MFMA GEMM / implicit-GEMM
// convolution then sums hi.Whi + hi.Wlo + lo.Whi in its fp32 accumulators: both operands carry ~22 mantissa bits, the
// dropped lo.Wlo term is 2^-22 of the result.  Each thread handles 8 consecutive channels of one token.
// ---------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void gn_apply_split3_kernel(const float* x, int C, int n, int64_t HW, int groups,
                                                              const float* stats, const float* gamma, const float* beta,
                                                              int silu, uint16_t* out, int64_t ldo, int c_off, int c_total) {
  // (x may be ONE source of a virtual channel concat of c_total channels starting at channel c_off: groups, gamma and beta
  // are those of the concatenated tensor)
  const int cpg = c_total / groups;
  const int c8 = C / 8;
  const int64_t total = (int64_t)n * HW * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t tok = i / c8;
    const int c = (int)(i - tok * c8) * 8;
    const int img = (int)(tok / HW);
    float v[8];
    const float4 a = *reinterpret_cast<const float4*>(x + tok * C + c);
    const float4 b = *reinterpret_cast<const float4*>(x + tok * C + c + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    if (stats) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int grp = (c_off + c + k) / cpg;
        const float mean = stats[((int64_t)img * groups + grp) * 2];
        const float rstd = stats[((int64_t)img * groups + grp) * 2 + 1];
        const float t = (v[k] - mean) * rstd * gamma[c_off + c + k] + beta[c_off + c + k];
        v[k] = silu ? silu_f(t) : t;
      }
    }
    float lo[8];
    const uint4 hi = pack8<DT>(v);
    float hf[8];
    unpack8<DT>(hi, hf);
#pragma unroll
    for (int k = 0; k < 8; ++k) lo[k] = v[k] - hf[k];
    uint16_t* o = out + tok * ldo + c;
    *reinterpret_cast<uint4*>(o) = hi;
    *reinterpret_cast<uint4*>(o + C) = hi;
    *reinterpret_cast<uint4*>(o + 2 * C) = pack8<DT>(lo);
  }
}

// GroupNorm folded to y = x * a + b per (image, channel): the operand of mimo_conv3x3_fused (hconv.hip)
__global__ __launch_bounds__(256) void gn_affine_kernel(const float* stats, const float* gamma, const float* beta, int n, int C,
                                                        int groups, float* ab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * C) return;
  const int img = i / C, c = i - img * C;
  const int grp = c / (C / groups);
  const float mean = stats[((int64_t)img * groups + grp) * 2], rstd = stats[((int64_t)img * groups + grp) * 2 + 1];
  const float a = rstd * gamma[c];
  ab[(int64_t)img * 2 * C + c] = a;
  ab[(int64_t)img * 2 * C + C + c] = fmaf(-mean, a, beta[c]);
}

// ---------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 64*8*4 = 2048).
// ---------------------------------------------------------------------------------
template <int DT, int VPL>  // VPL = 8-element vectors per lane
__global__ __launch_bounds__(256) void layer_norm_kernel(const void* x, int f32, int64_t rows, int C,
                                                         float eps, const float* gamma,
                                                         const float* beta, const float* pe,
                                                         int64_t rows_per_frame, int pe_frames,
                                                         uint16_t* out, float* out32) {
  const int lane = threadIdx.x & 63;
  const int c8 = C / 8;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  // affine parameters stay in registers for every row this wave normalises
  float gm[VPL][8], bt[VPL][8];
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int ci = lane + 64 * j;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gm[j][k] = ci < c8 ? gamma[ci * 8 + k] : 0.f;
      bt[j][k] = ci < c8 ? beta[ci * 8 + k] : 0.f;
    }
  }
  auto load_row = [&](int64_t row, float (&v)[VPL][8]) {
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ci = lane + 64 * j;
      if (ci < c8) {
        if (f32) {
          const float* p = (const float*)x + row * C + ci * 8;
          const float4 a = *reinterpret_cast<const float4*>(p);
          const float4 b = *reinterpret_cast<const float4*>(p + 4);
          v[j][0] = a.x; v[j][1] = a.y; v[j][2] = a.z; v[j][3] = a.w;
          v[j][4] = b.x; v[j][5] = b.y; v[j][6] = b.z; v[j][7] = b.w;
        } else {
          unpack8<DT>(*reinterpret_cast<const uint4*>((const uint16_t*)x + row * C + ci * 8), v[j]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[j][k] = 0.f;
      }
    }
  };
  float cur[VPL][8], nxt[VPL][8];
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  load_row(row, cur);
  for (; row < rows; row += nwaves) {
    const int64_t nrow = row + nwaves;
    if (nrow < rows) load_row(nrow, nxt);  // next row's loads fly under this row's reductions
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) s += cur[j][k];  // padded lanes hold zeros
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      if (lane + 64 * j < c8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = cur[j][k] - mean;
          q += d * d;
        }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    const float* pe_row = pe ? pe + ((row / rows_per_frame) % pe_frames) * (int64_t)C : nullptr;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
      const int ci = lane + 64 * j;
      if (ci < c8) {
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (cur[j][k] - mean) * rstd * gm[j][k] + bt[j][k];
        if (pe_row) {
          const float4 pa = *reinterpret_cast<const float4*>(pe_row + ci * 8);
          const float4 pb = *reinterpret_cast<const float4*>(pe_row + ci * 8 + 4);
          y[0] += pa.x; y[1] += pa.y; y[2] += pa.z; y[3] += pa.w;
          y[4] += pb.x; y[5] += pb.y; y[6] += pb.z; y[7] += pb.w;
        }
        if (out) *reinterpret_cast<uint4*>(out + row * C + ci * 8) = pack8<DT>(y);
        if (out32) {
          *reinterpret_cast<float4*>(out32 + row * C + ci * 8) = make_float4(y[0], y[1], y[2], y[3]);
          *reinterpret_cast<float4*>(out32 + row * C + ci * 8 + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) cur[j][k] = nxt[j][k];
  }
}

template <int DT>
__global__ __launch_bounds__(256) void cast_kernel(const void* in, int f32, int64_t count8, uint16_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    if (f32) {
      const float4 a = *reinterpret_cast<const float4*>((const float*)in + i * 8);
      const float4 b = *reinterpret_cast<const float4*>((const float*)in + i * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      unpack8<DT>(*reinterpret_cast<const uint4*>((const uint16_t*)in + i * 8), v);
    }
    *reinterpret_cast<uint4*>(out + i * 8) = pack8<DT>(v);
  }
}

inline unsigned stream_grid(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  // grid-stride over the rest; measured (tools/ln_micro.py --gn): the large tensors of levels 0 / 1 and of the VAE stream
  // 8-9 % faster from 16384 blocks than from 2048, the small ones prefer the short grid
  int64_t cap = work_items >= 6000000 ? 16384 : work_items >= 3000000 ? 4096 : 2048;
  if (const int64_t forced = tune_env("MIMO_STREAM_BLOCKS", 0)) cap = forced;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int mimo_group_norm_stats(const void* x1, int C1, const void* x2, int C2, int x_is_f32,
                                     int dtype, int n, int64_t HW, int groups, float eps, float* stats,
                                     float* partials, int split, void* stream) {
  if (split < 1 || (split > 1 && !partials) || split > HW) return MIMO_EINVAL;
  if (!x1 || !stats || n <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0) return MIMO_EINVAL;
  if ((C1 + C2) % groups) return MIMO_EINVAL;
  if (C2 > 0 && !x2) return MIMO_EINVAL;
  if (HW * ((C1 + C2) / groups) >= 0x7fffffffLL) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // row-streaming form: whenever the caller provides the partials buffer and the channel layout allows it
  const int C = C1 + C2;
  const bool rows_form = partials && !(C1 & 3) && !(C2 & 3) && (C >> 2) <= 1024 && groups <= (C >> 2) && split >= 1;
  if (rows_form) {
    if (dtype != MIMO_F16 && dtype != MIMO_BF16) return MIMO_EDTYPE;
    const int c4 = C >> 2;
    int R = 512 / c4;
    if (R < 1) R = 1;
    const unsigned threads = (unsigned)(c4 * R);
    const unsigned rgrid = (unsigned)(n * split);
    const size_t lds = (size_t)threads * 8 * sizeof(float);
    if (dtype == MIMO_F16)
      hipLaunchKernelGGL(gn_stats_rows_kernel<MIMO_F16>, dim3(rgrid), dim3(threads), lds, st, x1, C1, x2, C2, x_is_f32, HW, groups, partials, split);
    else
      hipLaunchKernelGGL(gn_stats_rows_kernel<MIMO_BF16>, dim3(rgrid), dim3(threads), lds, st, x1, C1, x2, C2, x_is_f32, HW, groups, partials, split);
    MIMO_LAUNCH_CHECK();
    const int total = n * groups;
    const unsigned fg = (unsigned)((total + 3) / 4);
    if (dtype == MIMO_F16)
      hipLaunchKernelGGL(gn_finalize_kernel<MIMO_F16>, dim3(fg), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, HW, groups, total, eps, partials, split, stats);
    else
      hipLaunchKernelGGL(gn_finalize_kernel<MIMO_BF16>, dim3(fg), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, HW, groups, total, eps, partials, split, stats);
    MIMO_LAUNCH_CHECK();
    return MIMO_OK;
  }
  const unsigned grid = (unsigned)(n * groups * split);
  const bool v2 = (((C1 + C2) / groups) % 2 == 0) && (C1 % 2 == 0) && (C2 % 2 == 0);
#define GNS_LAUNCH(DT, V) hipLaunchKernelGGL((gn_stats_kernel<DT, V>), dim3(grid), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, HW, groups, eps, stats, partials, split)
  if (dtype == MIMO_F16) {
    if (v2) GNS_LAUNCH(MIMO_F16, 2); else GNS_LAUNCH(MIMO_F16, 1);
  } else if (dtype == MIMO_BF16) {
    if (v2) GNS_LAUNCH(MIMO_BF16, 2); else GNS_LAUNCH(MIMO_BF16, 1);
  } else {
    return MIMO_EDTYPE;
  }
#undef GNS_LAUNCH
  MIMO_LAUNCH_CHECK();
  if (split > 1) {
    const int total = n * groups;
    const unsigned fg = (unsigned)((total + 3) / 4);
    if (dtype == MIMO_F16)
      hipLaunchKernelGGL(gn_finalize_kernel<MIMO_F16>, dim3(fg), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, HW, groups, total, eps, partials, split, stats);
    else
      hipLaunchKernelGGL(gn_finalize_kernel<MIMO_BF16>, dim3(fg), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, HW, groups, total, eps, partials, split, stats);
    MIMO_LAUNCH_CHECK();
  }
  return MIMO_OK;
}

extern "C" int mimo_group_norm_stats_slabs(const float* cs1, int C1, int rows_per_slab1, const float* cs2, int C2, int rows_per_slab2,
                                           int n, int64_t HW, int groups, float eps, float* stats, void* stream) {
  if (!cs1 || !stats || n <= 0 || HW <= 0 || rows_per_slab1 <= 0 || (HW % rows_per_slab1) || groups <= 0 || C1 <= 0 || C2 < 0) return MIMO_EINVAL;
  if ((C1 + C2) % groups || (C2 > 0 && (!cs2 || rows_per_slab2 <= 0 || (HW % rows_per_slab2)))) return MIMO_EINVAL;
  const int total = n * groups;
  hipLaunchKernelGGL(gn_stats_cols_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, cs1, C1,
                     rows_per_slab1, cs2, C2, C2 > 0 ? rows_per_slab2 : rows_per_slab1, HW, groups, eps, stats);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_group_norm_stats_cols(const float* cs1, int C1, const float* cs2, int C2, int n, int64_t HW,
                                          int groups, float eps, float* stats, void* stream) {
  return mimo_group_norm_stats_slabs(cs1, C1, 32, cs2, C2, 32, n, HW, groups, eps, stats, stream);
}

extern "C" int mimo_group_norm_apply(const void* x1, int C1, const void* x2, int C2, int x_is_f32,
                                     int dtype, int n, int64_t HW, int groups, const float* stats,
                                     const float* gamma, const float* beta, int silu, void* out,
                                     void* raw_out, void* stream) {
  if (!x1 || n <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0) return MIMO_EINVAL;
  if ((C1 & 7) || (C2 & 7) || (C1 + C2) % groups) return MIMO_EINVAL;
  if (C2 > 0 && !x2) return MIMO_EINVAL;
  if (out && (!stats || !gamma || !beta)) return MIMO_EINVAL;
  if (!out && !raw_out) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = stream_grid((int64_t)n * HW * ((C1 + C2) / 8));
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(gn_apply_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, n, HW, groups, stats, gamma, beta, silu, (uint16_t*)out, (uint16_t*)raw_out);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, x1, C1, x2, C2, x_is_f32, n, HW, groups, stats, gamma, beta, silu, (uint16_t*)out, (uint16_t*)raw_out);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_group_norm_apply_split3(const float* x, int C, int dtype, int n, int64_t HW, int groups, const float* stats,
                                            const float* gamma, const float* beta, int silu, void* out, int64_t ldo, int c_off,
                                            int c_total, void* stream) {
  if (!x || !out || n <= 0 || HW <= 0 || C <= 0 || (C & 7) || ldo < 3 * (int64_t)C || (ldo & 7)) return MIMO_EINVAL;
  if (c_total <= 0) { c_off = 0; c_total = C; }
  if (c_off < 0 || c_off + C > c_total) return MIMO_EINVAL;
  if (stats && (!gamma || !beta || groups <= 0 || c_total % groups)) return MIMO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u)) return MIMO_EINVAL;
  if (!stats) groups = 1;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = stream_grid((int64_t)n * HW * (C / 8));
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(gn_apply_split3_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, x, C, n, HW, groups, stats, gamma, beta, silu, (uint16_t*)out, ldo, c_off, c_total);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(gn_apply_split3_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, x, C, n, HW, groups, stats, gamma, beta, silu, (uint16_t*)out, ldo, c_off, c_total);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_group_norm_affine(const float* stats, const float* gamma, const float* beta, int n, int C, int groups,
                                      float* ab, void* stream) {
  if (!stats || !gamma || !beta || !ab || n <= 0 || C <= 0 || groups <= 0 || C % groups) return MIMO_EINVAL;
  hipLaunchKernelGGL(gn_affine_kernel, dim3((unsigned)((n * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta,
                     n, C, groups, ab);
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_layer_norm(const void* x, int x_is_f32, int dtype, int64_t rows, int C, float eps,
                               const float* gamma, const float* beta, const float* pe,
                               int64_t rows_per_frame, int pe_frames, void* out, float* out_f32, void* stream) {
  if (!x || (!out && !out_f32) || !gamma || !beta || rows <= 0 || C <= 0 || (C & 7) || C > 2048) return MIMO_EINVAL;
  if (pe && (rows_per_frame <= 0 || pe_frames <= 0)) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // Persistent waves, grid-stride over rows.  Few, long-lived waves win: a wave pays for its parameter loads and its first,
  // unhidden row once, and the bytes in flight (two rows per wave) only have to cover latency x bandwidth.  Measured optimum
  // (tools/ln_micro.py, profiles/r3_ln_grid_probe.txt): ~12 rows per wave, at most min(655360 / C, 1536) blocks — 12288 x 1280
  // takes 25.5 us with 512 blocks against 43.6 us with the former 2048.
  int64_t nb = (rows + 11) / 12;
  int64_t cap = 655360 / C < 1536 ? 655360 / C : 1536;
  if (const int64_t forced = tune_env("MIMO_LN_BLOCKS", 0)) nb = (rows + 3) / 4, cap = forced;
  if (cap < 1) cap = 1;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  const unsigned grid = (unsigned)nb;
  const int vpl = (C / 8 + 63) / 64;
#define LN_LAUNCH(DT, V)                                                                              \
  hipLaunchKernelGGL((layer_norm_kernel<DT, V>), dim3(grid), dim3(256), 0, st, x, x_is_f32, rows, C, \
                     eps, gamma, beta, pe, rows_per_frame, pe_frames, (uint16_t*)out, out_f32)
  if (dtype == MIMO_F16) {
    if (vpl == 1) LN_LAUNCH(MIMO_F16, 1); else if (vpl == 2) LN_LAUNCH(MIMO_F16, 2);
    else if (vpl == 3) LN_LAUNCH(MIMO_F16, 3); else LN_LAUNCH(MIMO_F16, 4);
  } else if (dtype == MIMO_BF16) {
    if (vpl == 1) LN_LAUNCH(MIMO_BF16, 1); else if (vpl == 2) LN_LAUNCH(MIMO_BF16, 2);
    else if (vpl == 3) LN_LAUNCH(MIMO_BF16, 3); else LN_LAUNCH(MIMO_BF16, 4);
  } else {
    return MIMO_EDTYPE;
  }
#undef LN_LAUNCH
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_cast(const void* in, int in_is_f32, int dtype, int64_t count, void* out_half,
                         void* stream) {
  if (!in || !out_half || count <= 0 || (count & 7)) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = stream_grid(count / 8);
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(cast_kernel<MIMO_F16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, count / 8, (uint16_t*)out_half);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(cast_kernel<MIMO_BF16>, dim3(grid), dim3(256), 0, st, in, in_is_f32, count / 8, (uint16_t*)out_half);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}
