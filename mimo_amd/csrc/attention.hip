// attention.hip — attention cores of the denoising path (gfx950, MFMA 32x32x16).
//
//  * attn_kernel: spatial multi-head flash attention with an optional second key/value
//    segment (the reference-attention bank) shared by all batch rows >= seg2_first_batch.
//    Replaces diffusers Attention/SDPA as driven by src/models/mutual_self_attention.py:154-197
//    (read mode: cond rows attend [self || bank], uncond rows attend self only — computed in
//    ONE launch, no overwrite pass) and :137-147 (write mode / plain self-attention).
//  * temporal_attn_kernel: attention over the <=32 frames of each (pixel, head), reading the
//    frame-major token layout in place (no '(b f) d c -> (b d) f c' copies).
//    Replaces VersatileAttention.forward, src/models/motion_module.py:353-390.
//  * softmax_rows_kernel: plain row softmax for the single-head d=512 VAE attention.
//
// Formulation ("swapped"): S^T = K.Q^T and O^T = V^T.P^T, so a lane owns ONE query column
// (q = lane & 31): the online-softmax max/sum are in-lane reductions plus one xor-32
// shuffle, the O rescale needs no broadcast, and P feeds the second MFMA straight from the
// accumulator registers (the k-index permutation is applied identically to V^T's fragment).
#include "common.cuh"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KV_TILE = 64;
constexpr float LOG2E = 1.4426950408889634f;

// raw v_exp_f32: arguments are <= 0 here, results below 2^-126 may flush to zero (libm exp2f adds a
// denormal-range rescale = 4 extra VALU ops per element, which matters in the softmax inner loop)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnArgs {
  const uint16_t *q, *k, *v, *k2, *v2;
  uint16_t* out;
  int64_t ldq, ldk, ldv, ldk2, ldv2, ldo;
  int B, Nq, Nk, Nk2, seg2_first_batch, heads;
  float scale_log2;
  int prescaled;  // Q already carries softmax_scale * log2(e) (folded into W_q by the caller)
};

template <int DT, int D, bool FAST>
__global__ __launch_bounds__(256, (D <= 64 ? 4 : 2)) void attn_kernel(const AttnArgs a) {
  constexpr int KS = (D + 15) / 16;        // QK^T k-steps of 16
  constexpr int OT = (D + 31) / 32;        // 32-row tiles of O^T
  constexpr int KP = KS * 16 + 8;          // K tile pitch (halfs): odd multiple of 16 bytes
  constexpr int VP = KV_TILE + 4;          // V^T tile pitch (halfs): 8 * odd bytes
  constexpr int DC = D / 8;                // 16-byte chunks per head row
  constexpr int NBUF = D <= 80 ? 2 : 1;   // double-buffered K / V^T tiles where LDS and registers allow it
  constexpr int KSZ = KV_TILE * KP, VSZ = OT * 32 * VP;
  __shared__ __attribute__((aligned(16))) uint16_t Ks[NBUF * KSZ];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[NBUF * VSZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, li = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;

  // zero the LDS once: pad columns / rows are never written again
  for (int i = tid; i < NBUF * KSZ / 2; i += 256) reinterpret_cast<uint32_t*>(Ks)[i] = 0u;
  for (int i = tid; i < NBUF * VSZ / 2; i += 256) reinterpret_cast<uint32_t*>(Vt)[i] = 0u;
  // When the O^T tiles have spare rows (D not a multiple of 32) row D of V^T is set to ones: the P.V MFMAs then
  // also produce the softmax denominator sum_kv P (in the O^T accumulator row D) — no per-element VALU adds.
  constexpr bool ONES = OT * 32 > D;
  // Spare K column D (when D is not a multiple of 16) is set to ones as well: with Q^T row D = -m the QK^T MFMAs
  // deliver S - m directly and the softmax needs no per-score subtract (fast path below).
  constexpr bool BIASCOL = KS * 16 > D;
  constexpr bool fast = FAST;
  static_assert(!FAST || BIASCOL, "the fast softmax path needs a spare K column");
  if (ONES || BIASCOL) __syncthreads();
  if (ONES)
    for (int i = tid; i < NBUF * VP; i += 256) Vt[(i / VP) * VSZ + D * VP + (i % VP)] = HT<DT>::from_f(1.0f);
  if (BIASCOL && fast)
    for (int i = tid; i < NBUF * KV_TILE; i += 256) Ks[(i / KV_TILE) * KSZ + (i % KV_TILE) * KP + D] = HT<DT>::from_f(1.0f);

  // Q^T fragments (B operand): lane (h2, q = li) holds Q[q][16 s + 8 h2 .. +8]
  uint4 qf[KS];
  {
    const int qr = q0 + li;
    const uint16_t* qb = a.q + (int64_t)b * a.Nq * a.ldq;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (int)((int64_t)a.Nq * a.ldq * 2), 0x00020000);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 16 * s + 8 * h2;
      const bool ok = (qr < a.Nq) & (kk < D);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? (unsigned)(((int64_t)qr * a.ldq + head * D + kk) * 2) : 0xFFFFFFF0u, 0, 0);
      qf[s] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }

  f32x16 ot[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // running max in RAW score units (the scale is folded into the exp2 fma)
  float mq = 0.f;                         // fast path: integer reference max currently folded into Q^T row D
  const float c = a.scale_log2;

  // flattened KV-tile list: the self segment, then (cond rows only) the bank segment
  const bool has2 = a.k2 && a.Nk2 > 0 && b >= a.seg2_first_batch;
  const int T0 = (a.Nk + KV_TILE - 1) / KV_TILE;
  const int T = T0 + (has2 ? (a.Nk2 + KV_TILE - 1) / KV_TILE : 0);
  const uint16_t* kb0 = a.k + (int64_t)b * a.Nk * a.ldk;
  const uint16_t* vb0 = a.v + (int64_t)b * a.Nk * a.ldv;
  constexpr int NCH = (KV_TILE * DC + 255) / 256;
  u32x4 kreg[NCH], vreg[NCH];

  // Per-thread staging geometry is loop-invariant: chunk `it` of a thread is (K row krow, 16-byte chunk kcc) and
  // (V row vrow, chunk vcc).  Only the tile's first KV row moves, and it travels in the scalar soffset.
  int krow_[NCH], vrow_[NCH];
  unsigned kcol_[NCH], vcol_[NCH], klds_[NCH], vlds_[NCH];
  bool live_[NCH];
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int id = tid + 256 * it;
    live_[it] = id < KV_TILE * DC;
    krow_[it] = id / DC;
    const int kcc = id - krow_[it] * DC;
    vrow_[it] = id & (KV_TILE - 1);
    const int vcc = id >> 6;
    kcol_[it] = (unsigned)((head * D + kcc * 8) * 2);
    vcol_[it] = (unsigned)((head * D + vcc * 8) * 2);
    klds_[it] = (unsigned)(krow_[it] * KP + kcc * 8);
    vlds_[it] = (unsigned)(vcc * 8 * VP + vrow_[it]);
  }
  constexpr unsigned OOBA = 0x80000000u;  // stays out of range after the (< 2 GiB) soffset is added

  // issue every global load of tile t (buffer loads; rows past the segment end read zeros)
  auto issue = [&](int t) {
    const bool s2 = t >= T0;
    const uint16_t* kb = s2 ? a.k2 : kb0;
    const uint16_t* vb = s2 ? a.v2 : vb0;
    const unsigned ldk2b = (unsigned)((s2 ? a.ldk2 : a.ldk) * 2), ldv2b = (unsigned)((s2 ? a.ldv2 : a.ldv) * 2);
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)((int64_t)nk * ldk2b), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)((int64_t)nk * ldv2b), 0x00020000);
    const unsigned ksoff = (unsigned)kv0 * ldk2b, vsoff = (unsigned)kv0 * ldv2b;
    const bool ragged = kv0 + KV_TILE > nk;  // wave-uniform: only the last tile of a segment pays for row checks
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      unsigned ko = (unsigned)krow_[it] * ldk2b + kcol_[it];
      unsigned vo = (unsigned)vrow_[it] * ldv2b + vcol_[it];
      bool kok = live_[it], vok = live_[it];
      if (ragged) {
        kok &= kv0 + krow_[it] < nk;
        vok &= kv0 + vrow_[it] < nk;
      }
      kreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rk, kok ? ko : OOBA, ksoff, 0);
      vreg[it] = __builtin_amdgcn_raw_buffer_load_b128(rv, vok ? vo : OOBA, vsoff, 0);
    }
  };

  // registers -> LDS buffer `buf`: K row-major, V transposed
  auto stage = [&](int buf) {
    uint16_t* ks = Ks + buf * KSZ;
    uint16_t* vt = Vt + buf * VSZ;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      if (live_[it]) {
        *reinterpret_cast<uint4*>(&ks[klds_[it]]) = make_uint4(kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w);
        const uint32_t w[4] = {vreg[it].x, vreg[it].y, vreg[it].z, vreg[it].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vt[vlds_[it] + (2 * i) * VP] = (uint16_t)(w[i] & 0xffffu);
          vt[vlds_[it] + (2 * i + 1) * VP] = (uint16_t)(w[i] >> 16);
        }
      }
    }
  };

  // Pipeline (NBUF = 2): ONE barrier per KV tile.  While tile t is consumed from buffer t&1, tile t+1 (already in
  // registers) is written to the other buffer and tile t+2's global loads are issued.
  // (NBUF = 1: barrier, stage, barrier, prefetch, consume.)
  issue(0);
  __syncthreads();  // orders the zero-fill
  if (NBUF == 2) {
    stage(0);
    if (T > 1) issue(1);
    __syncthreads();
  }
  for (int t = 0; t < T; ++t) {
    const int cur = NBUF == 2 ? (t & 1) : 0;
    if (NBUF == 1) {
      if (t > 0) __syncthreads();  // tile t-1 fully consumed
      stage(0);
      __syncthreads();
      if (t + 1 < T) issue(t + 1);
    }
    const uint16_t* ks = Ks + cur * KSZ;
    const uint16_t* vt = Vt + cur * VSZ;

    const bool s2 = t >= T0;
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;

    // ---- S^T = K.Q^T : two 32-kv sub-tiles ----
    f32x16 st[2];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint4 kf = *reinterpret_cast<const uint4*>(&ks[(32 * u + li) * KP + 16 * s + 8 * h2]);
        st[u] = HT<DT>::mfma32(kf, qf[s], s == 0 ? zero16 : st[u]);  // C = 0 folds into the instruction
      }
    }
    // ---- online softmax over kv for this lane's query (raw-score max; scale folded into the exp2 fma) ----
    if (kv0 + KV_TILE > nk) {  // ragged last tile of a segment (wave-uniform)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (kv >= nk) st[u][r] = -INFINITY;
        }
    }
    float mt = st[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[u][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    if (fast) {
      // st already is (score in log2 units) - mq, mq = this query's INTEGER reference max carried in Q^T row D
      // (exact in half precision and in the MFMA accumulate).  Any reference within THR of the true running max
      // is exact softmax arithmetic (p <= 2^THR, sums in fp32), so it only moves when a tile overshoots it.
      constexpr float THR = 6.f;
      const bool need = (t == 0) | (mt > THR);
      if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
        float dlt = need ? ceilf(mt) : 0.f;
        dlt = fminf(fmaxf(dlt, -2000.f - mq), 2000.f - mq);
        // the reference actually used is the half-precision value stored in Q^T (fp16: every integer up to 2048;
        // bf16: integers up to 256, coarser above) — move by the difference of the STORED values
        dlt = HT<DT>::to_f(HT<DT>::from_f(mq + dlt)) - mq;
        if (t != 0) {
          const float alpha = fast_exp2(-dlt);
#pragma unroll
          for (int dt = 0; dt < OT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
          if (!ONES) l_run *= alpha;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[u][r] -= dlt;
        mq += dlt;
        if (h2 == (D & 15) / 8) {  // the lane half whose fragment holds Q^T row D
          constexpr int E = D & 7;
          uint32_t* w = &qf[KS - 1].x + E / 2;
          const uint32_t hb = HT<DT>::from_f(-mq);
          *w = (E & 1) ? ((*w & 0x0000ffffu) | (hb << 16)) : ((*w & 0xffff0000u) | hb);
        }
      }
      float ps = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(st[u][r]);
          st[u][r] = pv;
          if (!ONES) ps += pv;
        }
      if (!ONES) {
        ps += __shfl_xor(ps, 32, 64);
        l_run += ps;
      }
    } else {
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {  // some lane's max moved: rescale (rare after a few tiles)
      const float alpha = fast_exp2((m_run - m_use) * c);  // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < OT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[dt][r] *= alpha;
    }
    m_run = m_new;
    const float mc = -m_use * c;
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = fast_exp2(fmaf(st[u][r], c, mc));
        st[u][r] = pv;
        if (!ONES) ps += pv;
      }
    if (!ONES) {
      ps += __shfl_xor(ps, 32, 64);
      l_run += ps;
    }
    }

    // ---- P^T fragments (B operand) straight from the accumulators ----
    uint4 pf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = u >> 1, hh = u & 1;
      pf[u].x = pack2<DT>(st[tt][8 * hh + 0], st[tt][8 * hh + 1]);
      pf[u].y = pack2<DT>(st[tt][8 * hh + 2], st[tt][8 * hh + 3]);
      pf[u].z = pack2<DT>(st[tt][8 * hh + 4], st[tt][8 * hh + 5]);
      pf[u].w = pack2<DT>(st[tt][8 * hh + 6], st[tt][8 * hh + 7]);
    }
    // ---- O^T += V^T.P^T ----
#pragma unroll
    for (int dt = 0; dt < OT; ++dt) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint16_t* vr = &vt[(32 * dt + li) * VP + 16 * u + 4 * h2];
        const uint2 lo = *reinterpret_cast<const uint2*>(vr);
        const uint2 hi = *reinterpret_cast<const uint2*>(vr + 8);
        ot[dt] = HT<DT>::mfma32(make_uint4(lo.x, lo.y, hi.x, hi.y), pf[u], ot[dt]);
      }
    }
    if (NBUF == 2 && t + 1 < T) {
      stage(cur ^ 1);              // last read in iteration t-1, every wave has passed that iteration's barrier
      if (t + 2 < T) issue(t + 2);
      __syncthreads();             // publishes tile t+1; every wave is done with buffer `cur`
    }
  }

  // ---- epilogue: lane (h2, q) holds O^T[d = 32 dt + (r&3) + 8 (r>>2) + 4 h2][q] ----
  if (ONES) {
    // row D = 32*(OT-1) + (D & 31) of O^T: register r with (r&3) + 8*(r>>2) + 4*h2 == D & 31, held by one h2 half
    constexpr int RL = D & 31;
    constexpr int LH2 = (RL >> 2) & 1, LR = (RL & 3) + 4 * (RL >> 3);
    l_run = __shfl(ot[OT - 1][LR], li + 32 * LH2, 64);
  }
  const int qr = q0 + li;
  if (qr < a.Nq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    uint16_t* op = a.out + ((int64_t)b * a.Nq + qr) * a.ldo + head * D;
#pragma unroll
    for (int dt = 0; dt < OT; ++dt)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = 32 * dt + 8 * c + 4 * h2;
        if (d < D) {
          uint2 o;
          o.x = pack2<DT>(ot[dt][4 * c + 0] * inv, ot[dt][4 * c + 1] * inv);
          o.y = pack2<DT>(ot[dt][4 * c + 2] * inv, ot[dt][4 * c + 3] * inv);
          *reinterpret_cast<uint2*>(op + d) = o;
        }
      }
  }
}

// ------------------------------------------------------------------------------------
// Temporal attention: one wave per (batch b, pixel p, head h); sequence = F <= 32 frames.
// ------------------------------------------------------------------------------------
struct TAttnArgs {
  const uint16_t *q, *k, *v;
  uint16_t* out;
  int64_t ldq, ldk, ldv, ldo, HW;
  int b, F, heads;
  float scale_log2;
};

template <int DT, int D>
__global__ __launch_bounds__(256, 2) void temporal_attn_kernel(const TAttnArgs a) {
  constexpr int KS = (D + 15) / 16;
  constexpr int OT = (D + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, li = lane & 31;
  // unit id -> (b, pixel, head); heads fastest so the 4 waves of a block share token rows
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nunits = (int64_t)a.b * a.HW * a.heads;
  if (unit >= nunits) return;
  const int head = (int)(unit % a.heads);
  const int64_t bp = unit / a.heads;
  const int64_t pix = bp % a.HW;
  const int bi = (int)(bp / a.HW);
  const int F = a.F;
  // token row of frame f: (bi*F + f)*HW + pix
  const int64_t row0 = (int64_t)bi * F * a.HW + pix;

  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
  // every global read goes through a buffer descriptor: frames >= F / channels >= D use an out-of-range offset
  // and read zeros, so all loads issue back-to-back without branches
  const int64_t tot_rows = (int64_t)a.b * F * a.HW;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (int)(tot_rows * a.ldq * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)a.k, 0, (int)(tot_rows * a.ldk * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)a.v, 0, (int)(tot_rows * a.ldv * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  {
    const bool fok = li < F;
    const int64_t row = row0 + (int64_t)li * a.HW;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 16 * s + 8 * h2;
      const bool ok = fok & (kk < D);
      const u32x4 kv4 = __builtin_amdgcn_raw_buffer_load_b128(rk, ok ? (unsigned)((row * a.ldk + head * D + kk) * 2) : OOB, 0, 0);
      const u32x4 qv4 = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? (unsigned)((row * a.ldq + head * D + kk) * 2) : OOB, 0, 0);
      st = HT<DT>::mfma32(make_uint4(kv4.x, kv4.y, kv4.z, kv4.w), make_uint4(qv4.x, qv4.y, qv4.z, qv4.w), st);
    }
  }
  float mt = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kv = (r & 3) + 8 * (r >> 2) + 4 * h2;
    const float sv = kv < F ? st[r] * a.scale_log2 : -INFINITY;
    st[r] = sv;
    mt = fmaxf(mt, sv);
  }
  mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
  float ps = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float p = fast_exp2(st[r] - mt);
    st[r] = p;
    ps += p;
  }
  ps += __shfl_xor(ps, 32, 64);
  const float inv = 1.f / ps;
  uint4 pf[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    pf[u].x = pack2<DT>(st[8 * u + 0], st[8 * u + 1]);
    pf[u].y = pack2<DT>(st[8 * u + 2], st[8 * u + 3]);
    pf[u].z = pack2<DT>(st[8 * u + 4], st[8 * u + 5]);
    pf[u].w = pack2<DT>(st[8 * u + 6], st[8 * u + 7]);
  }
#pragma unroll
  for (int dt = 0; dt < OT; ++dt) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const int d = 32 * dt + li;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      uint16_t e[8];
      // lane part of the address in the voffset, (compile-time) frame part in the scalar soffset
      const unsigned vlane = (unsigned)(((row0 + (int64_t)(4 * h2) * a.HW) * a.ldv + head * D + d) * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int fu = 16 * u + (j & 3) + 8 * (j >> 2);  // frame = fu + 4 h2
        const bool ok = (d < D) & (fu + 4 * h2 < F);
        e[j] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rv, ok ? vlane : OOB, (unsigned)((int64_t)fu * a.HW * a.ldv * 2), 0);
      }
      uint4 vf;
      vf.x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
      vf.y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
      vf.z = (uint32_t)e[4] | ((uint32_t)e[5] << 16);
      vf.w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
      o = HT<DT>::mfma32(vf, pf[u], o);
    }
    // lane (h2, q = li): o[r] = O^T[d = 32 dt + (r&3) + 8 (r>>2) + 4 h2][q]
    if (li < F) {
      uint16_t* op = a.out + (row0 + (int64_t)li * a.HW) * a.ldo + head * D;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = 32 * dt + 8 * c + 4 * h2;
        if (dd < D) {
          uint2 w;
          w.x = pack2<DT>(o[4 * c + 0] * inv, o[4 * c + 1] * inv);
          w.y = pack2<DT>(o[4 * c + 2] * inv, o[4 * c + 3] * inv);
          *reinterpret_cast<uint2*>(op + dd) = w;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Row softmax (fp32 in, half out): one block per row.
// ------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* in, int64_t ldi, uint16_t* out,
                                                           int64_t ldo, int cols, float scale_log2) {
  const int64_t row = blockIdx.x;
  const float* x = in + row * ldi;
  __shared__ float red[8];
  float m = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) m = fmaxf(m, x[c] * scale_log2);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += fast_exp2(x[c] * scale_log2 - m);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = threadIdx.x; c < cols; c += 256)
    out[row * ldo + c] = HT<DT>::from_f(fast_exp2(x[c] * scale_log2 - m) * inv);
}

}  // namespace

extern "C" int mimo_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, const void* k2, int64_t ldk2, const void* v2,
                              int64_t ldv2, void* out, int64_t ldo, int B, int Nq, int Nk, int Nk2,
                              int seg2_first_batch, int heads, int d, float scale, void* stream) {
  if (!q || !k || !v || !out || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0) return MIMO_EINVAL;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return MIMO_EINVAL;
  if (Nk2 > 0 && (!k2 || !v2 || (ldk2 & 7) || (ldv2 & 7))) return MIMO_EINVAL;
  if (heads > 65535 || B > 65535) return MIMO_EINVAL;
  AttnArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v;
  a.k2 = Nk2 > 0 ? (const uint16_t*)k2 : nullptr; a.v2 = Nk2 > 0 ? (const uint16_t*)v2 : nullptr;
  a.out = (uint16_t*)out;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldk2 = ldk2; a.ldv2 = ldv2; a.ldo = ldo;
  a.B = B; a.Nq = Nq; a.Nk = Nk; a.Nk2 = Nk2; a.seg2_first_batch = seg2_first_batch; a.heads = heads;
  a.scale_log2 = scale > 0.f ? scale * LOG2E : 1.0f;
  a.prescaled = scale > 0.f ? 0 : 1;  // scale <= 0: Q is pre-multiplied by softmax_scale * log2(e)
  const dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)heads, (unsigned)B);
  hipStream_t st = (hipStream_t)stream;
#define ATTN_LAUNCH(DT, DD) hipLaunchKernelGGL((attn_kernel<DT, DD, false>), grid, dim3(256), 0, st, a)
#define ATTN_LAUNCH40(DT)                                                                          \
  do {                                                                                             \
    if (a.prescaled) hipLaunchKernelGGL((attn_kernel<DT, 40, true>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((attn_kernel<DT, 40, false>), grid, dim3(256), 0, st, a);              \
  } while (0)
  if (dtype == MIMO_F16) {
    switch (d) {
      case 40: ATTN_LAUNCH40(MIMO_F16); break;
      case 64: ATTN_LAUNCH(MIMO_F16, 64); break;
      case 80: ATTN_LAUNCH(MIMO_F16, 80); break;
      case 160: ATTN_LAUNCH(MIMO_F16, 160); break;
      default: return MIMO_EINVAL;
    }
  } else if (dtype == MIMO_BF16) {
    switch (d) {
      case 40: ATTN_LAUNCH40(MIMO_BF16); break;
      case 64: ATTN_LAUNCH(MIMO_BF16, 64); break;
      case 80: ATTN_LAUNCH(MIMO_BF16, 80); break;
      case 160: ATTN_LAUNCH(MIMO_BF16, 160); break;
      default: return MIMO_EINVAL;
    }
  } else {
    return MIMO_EDTYPE;
  }
#undef ATTN_LAUNCH
#undef ATTN_LAUNCH40
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_temporal_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                       const void* v, int64_t ldv, void* out, int64_t ldo, int b, int F,
                                       int64_t HW, int heads, int d, float scale, void* stream) {
  if (!q || !k || !v || !out || b <= 0 || F <= 0 || F > 32 || HW <= 0 || heads <= 0) return MIMO_EINVAL;
  if ((ldq & 7) || (ldk & 7) || (ldo & 3)) return MIMO_EINVAL;
  TAttnArgs a;
  a.q = (const uint16_t*)q; a.k = (const uint16_t*)k; a.v = (const uint16_t*)v; a.out = (uint16_t*)out;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.HW = HW; a.b = b; a.F = F; a.heads = heads;
  a.scale_log2 = scale * LOG2E;
  const int64_t units = (int64_t)b * HW * heads;
  const int64_t nb = (units + 3) / 4;
  if (nb > 0x7fffffff) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#define TATTN_LAUNCH(DT, DD) hipLaunchKernelGGL((temporal_attn_kernel<DT, DD>), dim3((unsigned)nb), dim3(256), 0, st, a)
  if (dtype == MIMO_F16) {
    switch (d) {
      case 40: TATTN_LAUNCH(MIMO_F16, 40); break;
      case 64: TATTN_LAUNCH(MIMO_F16, 64); break;
      case 80: TATTN_LAUNCH(MIMO_F16, 80); break;
      case 160: TATTN_LAUNCH(MIMO_F16, 160); break;
      default: return MIMO_EINVAL;
    }
  } else if (dtype == MIMO_BF16) {
    switch (d) {
      case 40: TATTN_LAUNCH(MIMO_BF16, 40); break;
      case 64: TATTN_LAUNCH(MIMO_BF16, 64); break;
      case 80: TATTN_LAUNCH(MIMO_BF16, 80); break;
      case 160: TATTN_LAUNCH(MIMO_BF16, 160); break;
      default: return MIMO_EINVAL;
    }
  } else {
    return MIMO_EDTYPE;
  }
#undef TATTN_LAUNCH
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_softmax_rows(int dtype, const float* in, int64_t ldi, void* out, int64_t ldo,
                                 int64_t rows, int cols, float scale, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || rows > 0x7fffffff) return MIMO_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIMO_F16)
    hipLaunchKernelGGL(softmax_rows_kernel<MIMO_F16>, dim3((unsigned)rows), dim3(256), 0, st, in, ldi, (uint16_t*)out, ldo, cols, scale * LOG2E);
  else if (dtype == MIMO_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<MIMO_BF16>, dim3((unsigned)rows), dim3(256), 0, st, in, ldi, (uint16_t*)out, ldo, cols, scale * LOG2E);
  else
    return MIMO_EDTYPE;
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}
