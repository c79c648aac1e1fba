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
#include "common.hip.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KV_TILE = 64;
constexpr float LOG2E = 1.4426950408889634f;
template <int V>
struct IC2 {
  static constexpr int value = V;
};

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

// QK8: the opt-in fp8 (e4m3, OCP) Q.K^T variant of BASELINE configs[4] ("fp8 MFMA attention QK", accuracy reported, not
// gated): the K tile is converted to fp8 once while it is staged into LDS, Q once per block, and S^T = K.Q^T runs on
// v_mfma_f32_32x32x16_fp8_fp8; softmax and P.V are unchanged (half operands).  On gfx950 the non-scaled fp8 MFMA has the
// SAME rate as the f16 one (MI355X_MICROARCH.md: only the MX-scaled K = 128 forms double it, and d = 40 would pad them
// to 31 % occupancy), so this buys LDS bytes, not MFMA time: it exists to measure the accuracy cost.
typedef long i64_t;
__device__ __forceinline__ i64_t half8_to_fp8(const uint4& v, int dt) {
  float f[8];
  if (dt == MIMO_F16) unpack8<MIMO_F16>(v, f);
  else unpack8<MIMO_BF16>(v, f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return (i64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}

template <int DT, int D, bool FAST, bool QK8 = false>
__global__ __launch_bounds__(256, (D <= 64 ? 4 : 2)) void attn_kernel(const AttnArgs a) {
  constexpr int KS = (D + 15) / 16;        // QK^T k-steps of 16
  constexpr int OT = (D + 31) / 32;        // 32-row tiles of O^T
  constexpr int KP = KS * 16 + 8;          // K tile pitch (halfs): odd multiple of 16 bytes
  constexpr int VP = KV_TILE + 4;          // V^T tile pitch (halfs): 8 * odd bytes
  constexpr int DC = D / 8;                // 16-byte chunks per head row
  constexpr int NBUF = D <= 80 ? 2 : 1;   // double-buffered K / V^T tiles where LDS and registers allow it
  constexpr int KP8 = KS * 16 + 8;         // QK8: K tile pitch in BYTES (fp8), 8 * odd
  constexpr int KSZ = QK8 ? (KV_TILE * KP8 + 1) / 2 : KV_TILE * KP, VSZ = OT * 32 * VP;
  static_assert(!(QK8 && FAST), "the fp8 variant uses the plain online softmax");
  __shared__ __attribute__((aligned(16))) uint16_t Ks[NBUF * KSZ];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[NBUF * VSZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, li = lane & 31;
  // Block -> (query block, head, batch).  Workgroups are handed to the 8 XCDs round-robin in linear dispatch order (x fastest),
  // so with the plain blockIdx mapping the query blocks of ONE (batch, head) — which all stream the same K / V — land on 8
  // different L2s and each of them fetches that K / V again (measured: 1.22 GB fetched per level-1 launch for 0.19 GB of
  // operands, profiles/r6_pmc_traffic_by_kernel_before_grouping.txt).  xcd_remap gives an XCD a contiguous run of logical ids:
  // consecutive query blocks of a head share one L2.
  const unsigned nqb_ = gridDim.x, nh_ = gridDim.y;
  const unsigned Lr_ = xcd_remap((blockIdx.z * nh_ + blockIdx.y) * nqb_ + blockIdx.x, nqb_ * nh_ * gridDim.z);
  const int head = (int)((Lr_ / nqb_) % nh_);
  int b = (int)(Lr_ / (nqb_ * nh_));
  // (a contiguous run per XCD would give the first XCDs the short, self-only rows of a CFG batch and the last ones the rows
  // that also attend the bank: alternate long and short batch rows, as attn40_kernel does)
  if (a.k2 && a.Nk2 > 0 && 2 * a.seg2_first_batch == a.B) b = (b & 1) ? (b >> 1) : a.seg2_first_batch + (b >> 1);
  const int q0 = (int)(Lr_ % nqb_) * 128 + wave * 32;

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

  i64_t qf8[KS];
  if (QK8) {
#pragma unroll
    for (int s = 0; s < KS; ++s) qf8[s] = half8_to_fp8(qf[s], DT);
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
        if (QK8) {  // 8 halfs -> 8 fp8 bytes at row * KP8 + 8 * chunk
          const int kcc8 = (int)(klds_[it] - (unsigned)krow_[it] * KP) / 8;
          *reinterpret_cast<i64_t*>(reinterpret_cast<unsigned char*>(ks) + krow_[it] * KP8 + kcc8 * 8) =
              half8_to_fp8(make_uint4(kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w), DT);
        } else {
          *reinterpret_cast<uint4*>(&ks[klds_[it]]) = make_uint4(kreg[it].x, kreg[it].y, kreg[it].z, kreg[it].w);
        }
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
        if (QK8) {
          const i64_t kf8 = *reinterpret_cast<const i64_t*>(reinterpret_cast<const unsigned char*>(ks) + (32 * u + li) * KP8 + 16 * s + 8 * h2);
          st[u] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kf8, qf8[s], s == 0 ? zero16 : st[u], 0, 0, 0);
        } else {
          const uint4 kf = *reinterpret_cast<const uint4*>(&ks[(32 * u + li) * KP + 16 * s + 8 * h2]);
          st[u] = HT<DT>::mfma32(kf, qf[s], s == 0 ? zero16 : st[u]);  // C = 0 folds into the instruction
        }
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
// attn40_kernel: the d = 40 level (N = 4096 at 512x512: the largest single kernel of the denoising forward),
// Q pre-multiplied by softmax_scale * log2(e).  Differences from attn_kernel, each aimed at a measured limit of it
// (profiles/r1_pmc_sq_counters_attn_and_conv.txt, profiles/r2_attn40_*.txt):
//   * a wave owns 64 queries (two 32-query blocks A, B), a block 256: every K / V fragment read from LDS feeds twice
//     the MFMAs, and the K/V stream per FLOP is half that of a 128-query block (fabric-side fetch 3.0 GB -> 0.7 GB);
//   * P.V runs on 16x16x32 MFMAs: O^T is 48 rows (40 + the ones-row that yields the softmax denominators), not 64.
//     The 32x32 S^T accumulators become 16x16x32 B operands with one v_permlane16_swap per packed register pair
//     (odd 16-lane rows of one register <-> even rows of the other), no LDS round trip;
//   * the softmax costs exp + pack only: the integer reference max rides in the spare QK^T k-slot (as in
//     attn_kernel<40, FAST>), and instead of a running max the packed P words are OR-ed together — bit 14 of a half
//     is set exactly when p >= 2, i.e. a score overshot the reference (which is kept 3 above the running max).
//     Only then (and on the first tile) S is recomputed and re-referenced: exact softmax arithmetic either way;
//   * BOTH K and V tiles go global -> LDS by DMA, row-major as they lie in memory (row pitch 80 B), into a 3-deep
//     ring: the loads of tile t+2 are issued before tile t is consumed and waited for with a COUNTED vmcnt (no VGPR
//     staging, no compiler-inserted vmcnt(0) drain; +5 % over the 2-buffer form).  Tune-build ablations
//     (profiles/r2_attn40_ab.txt): never waiting for the DMAs gains 3 %, register staging instead of DMA ties, K/V tiles
//     contiguous in memory tie, 8-wave blocks lose 12 % — the loop is issue-bound at the power-limited clock, not
//     memory-bound (L2 hit rate 95 %, fabric fetch 0.7 GB); the variant that re-reads ONE tile runs 27 % faster only
//     because constant operands let the chip clock higher.  V^T fragments come out of the row-major V tile with ds_read_b64_tr_b16 (hardware
//     transpose: lane (g, i) receives V[kv0 + j][d0 + i], j < 4); K column 40 (the bias column of the reference-max
//     trick) and V^T rows 40..47 (ones-row + padding) are constant LDS slots the fragment addresses point at;
//   * 1-D grid, XCD-aware: consecutive logical blocks (same batch row and head = same K/V) run on one XCD and hit in
//     its L2; cond (two KV segments) and uncond batch rows alternate so every XCD gets the same work.
// ------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));

// V tile row placement.  A 32-lane group of ds_read_b64_tr_b16 gathers 8 kv rows x 32 B: rows r..r+3 and r+16..r+19 (the k-slot
// order of the P operand).  Row-major at the 80-byte pitch of the lane-linear DMA image, rows k and 16 + k fall on the same
// banks (16 x 80 = 5 x 256 B) and row 3 wraps onto row 0: every transposed read took >= 2 LDS cycles per group
// (SQ_LDS_BANK_CONFLICT = 55 % of the kernel's LDS-array cycles, profiles/r3_pmc_lds_by_family_final.txt).  The 8 spans tile
// the 256-byte bank window exactly when the rows sit in 8 slots of equal parity inside one 16-slot window (slot s starts at
// 16-byte unit 5 s mod 16), so a tile row r goes to slot
//     32 (r / 32) + 16 b8 + 8 b16 + 2 (r & 3) + b4        (b4, b8, b16 = bits 2, 3, 4 of r)
// — a bit permutation, applied on the SOURCE row of the DMA lanes; the K tile keeps its order (its ds_read_b128 groups are
// conflict-free at this pitch).
#ifndef MIMO_ATTN40_VPERM
#define MIMO_ATTN40_VPERM 1
#endif
__device__ __forceinline__ int v_slot_of_row(int r) {
  return MIMO_ATTN40_VPERM ? (r & 32) | (((r >> 3) & 1) << 4) | (((r >> 4) & 1) << 3) | ((r & 3) << 1) | ((r >> 2) & 1) : r;
}
__device__ __forceinline__ int v_row_of_slot(int s) {
  return MIMO_ATTN40_VPERM ? (s & 32) | (((s >> 3) & 1) << 4) | (((s >> 4) & 1) << 3) | ((s & 1) << 2) | ((s >> 1) & 3) : s;
}

// ABL (tune build only): 1 = no global traffic after the prologue (every tile recomputes on the first tiles' K / V):
// the compute-only time of the loop; 2 = DMAs issued but never waited for (stale tiles): issue cost without wait cost
// NW = waves per block (4 | 8): a block covers 64 NW queries.  The per-tile DMA work (10 instructions, ~100+ issue cycles
// each inside a busy phase) is fixed per block, so 8 waves halve its cost per FLOP.
// STAGE: 0 = tiles by LDS-DMA (asm, counted vmcnt); 1 = tiles through registers: plain buffer loads issued at the top of the
// iteration two tiles ahead, ds_write_b128 into the ring at its end (loads the compiler counts itself; NW = 4 only)
// PRIO: 0 = no priority hints; 1 = s_setprio(1) around both MFMA clusters of a tile (Q.K^T and P.V); 2 = around P.V only.
// Two blocks share a CU, so each SIMD hosts two waves of DIFFERENT blocks whose phases drift: the hint lets the wave that
// is entering an MFMA cluster win the issue arbitration against its partner's exp / pack VALU stream (cdna_hip_programming
// T5; measured in profiles/r3_attn40_setprio_ab.txt).
// PIPE (round 5): software pipeline INSIDE the wave.  A wave issues in order: with the per-tile sequence Q.K^T MFMAs -> exp / pack
// VALU -> P.V MFMAs the matrix pipe idles while the wave exponentiates, and the SIMD partner (a wave of the other resident
// block) drifts into the same phase — measured MFMA-busy 0.48 = the serial sum 768 matrix + ~700 VALU cycles per wave and tile
// (profiles/r5_attn40_isa_mix.txt, r5_mfma_ceiling.txt: v_exp_f32 6 cycles, other VALU 4 per wave64 instruction).  PIPE = 1
// computes the scores of tile t + 1 while tile t is exponentiated: the Q.K^T MFMAs of t + 1 and the exp / pack of t are
// independent and interleaved in program order (sched_group_barrier: one MFMA, then its share of the VALU work), P.V of query
// block A runs under the last third of the exponentials; the ring is 4 deep (K of t + 1 is read one tile earlier).
template <int DT, int NW, int ABL = 0, int STAGE = 0, int PRIO = 0, int PIPE = 0>
__global__ __launch_bounds__(64 * NW, PIPE == 2 ? 1 : 2) void attn40_kernel(const AttnArgs a) {
  constexpr int D = 40;
  constexpr int NB = (PIPE == 1 || PIPE == 2) ? 4 : 3;   // ring depth (tiles t, t+1, t+2 [, t+3])
  constexpr int ROWB = 80;                  // K / V tile row pitch in bytes (= the row itself: the DMA image is lane-linear)
  constexpr int TILEB = KV_TILE * ROWB;     // 5120 B per tile
  constexpr int K_OFF = 0, V_OFF = NB * TILEB;
  constexpr int CONST_OFF = 2 * NB * TILEB; // [0,16): halfs (1, 0 x 7) = K columns 40..47; [16,24): (1,0,0,0) = V^T row 40 of 4 kv; [24,32): zeros
  constexpr float HEAD = 3.f;               // the reference sits HEAD above the running max: p <= 2^-3 until it is overshot
  // ONE LDS object (a second one would make hipcc drain vmcnt before every fragment read)
  __shared__ __attribute__((aligned(16))) unsigned char smem[CONST_OFF + 64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, li = lane & 31;   // 32x32 MFMA view: query column li, row half h2
  const int g = lane >> 4, i16 = lane & 15;   // 16x16 MFMA view: k-slot group g, row / column i16

  // ---- block -> (batch row, head, 256-query block); logical id L walks query blocks fastest ----
  constexpr int QB = 64 * NW;
  const unsigned nqb = (unsigned)((a.Nq + QB - 1) / QB);
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned qblk = L % nqb;
  const unsigned hb = L / nqb;
  const int head = (int)(hb % (unsigned)a.heads);
  int b = (int)(hb / (unsigned)a.heads);
  if (a.k2 && a.Nk2 > 0 && 2 * a.seg2_first_batch == a.B)  // alternate long (two-segment) and short batch rows
    b = (b & 1) ? (b >> 1) : a.seg2_first_batch + (b >> 1);
  const int q0 = (int)qblk * QB + wave * 64;

  if (tid < 8) reinterpret_cast<uint32_t*>(smem + CONST_OFF)[tid] = (tid == 0 || tid == 4) ? (uint32_t)HT<DT>::from_f(1.0f) : 0u;

  // ---- Q^T fragments (B operand of the 32x32x16 MFMA): lane (h2, q = li) holds Q[q][16 s + 8 h2 .. +8] ----
  uint4 qf[2][3];
  {
    const uint16_t* qb = a.q + (int64_t)b * a.Nq * a.ldq;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (int)((int64_t)a.Nq * a.ldq * 2), 0x00020000);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int qr = q0 + 32 * x + li;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int kk = 16 * s + 8 * h2;
        const bool ok = (qr < a.Nq) & (kk < D);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? (unsigned)(((int64_t)qr * a.ldq + head * D + kk) * 2) : 0xFFFFFFF0u, 0, 0);
        qf[x][s] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the Q loads are the only compiler-tracked loads: the DMA counts below start at zero

  // O^T accumulators, 16x16 tiles: ot[x][dt][j]: lane (g, n) holds O^T[d = 16 dt + 4 g + r][q = 32 x + 16 j + n]
  f32x4 ot[2][3][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
      for (int z = 0; z < 2; ++z) ot[x][y][z] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mq[2] = {0.f, 0.f};  // integer reference folded into Q^T row 40 of each query block

  const bool has2 = a.k2 && a.Nk2 > 0 && b >= a.seg2_first_batch;
  const int T0 = (a.Nk + KV_TILE - 1) / KV_TILE;
  const int T = T0 + (has2 ? (a.Nk2 + KV_TILE - 1) / KV_TILE : 0);
  const uint16_t* kb0 = a.k + (int64_t)b * a.Nk * a.ldk;
  const uint16_t* vb0 = a.v + (int64_t)b * a.Nk * a.ldv;

  // ---- tile DMA: K and V are 320 16-byte chunks each = 5 + 5 wave-DMAs per tile.  Wave w issues K piece w, V piece w and
  // (w = 0) K piece 4 / (w = 1) V piece 4: 3 | 3 | 2 | 2 DMAs per wave (NW = 4; with NW = 8 waves 4..7 take the V pieces and
  // waves 0 / 4 the fifth pieces: 2 | 1 | 1 | 1 | 2 | 1 | 1 | 1) — the counts the vmcnt waits below rely on.
  // The no-wait ablation showed the loop pays for ISSUING these DMAs (instructions), not for waiting on them: everything
  // lane-dependent is loop-invariant (voff*), the tile's first row travels in the scalar offset, descriptors are picked by
  // scalar selects, and only a ragged last tile computes row masks. ----
  auto make_rsrc = [](const void* ptr, unsigned bytes) -> i32x4 {
    const uint64_t p64 = reinterpret_cast<uint64_t>(ptr);
    i32x4 r;
    r.x = (int)(uint32_t)p64; r.y = (int)((uint32_t)(p64 >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
  };
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&smem[0];
  // piece p (0..4) of a tile = chunks 64 p .. 64 p + 63; this lane's chunk: row c / 5, 16-byte column c % 5
  constexpr int NJ = NW == 4 ? 3 : 2;       // DMAs of the busiest wave
  const int pA = NW == 4 ? wave : (wave & 3);               // piece of this wave's first DMA ...
  const bool vA = NW == 4 ? false : wave >= 4;              // ... which is a K piece (NW = 4) or K / V by wave half (NW = 8)
  const bool has3 = NW == 4 ? wave < 2 : (wave & 3) == 0;   // this wave also moves a fifth piece
  const bool v3 = NW == 4 ? wave == 1 : wave == 4;          // ... of V (else of K)
  const unsigned ldk1 = (unsigned)(a.ldk * 2), ldv1 = (unsigned)(a.ldv * 2), ldk2 = (unsigned)(a.ldk2 * 2), ldv2 = (unsigned)(a.ldv2 * 2);
  int prow[2];
  unsigned pcol[2];
  {
    const int c0 = 64 * pA + lane, c1 = 256 + lane;
    prow[0] = c0 / 5; pcol[0] = (unsigned)((head * D + (c0 - prow[0] * 5) * 8) * 2);
    prow[1] = c1 / 5; pcol[1] = (unsigned)((head * D + (c1 - prow[1] * 5) * 8) * 2);
  }
  // byte offsets of this lane's chunks relative to the tile's first row, per segment (self | bank)
  const unsigned voffK[2] = {(unsigned)prow[0] * ldk1 + pcol[0], (unsigned)prow[0] * ldk2 + pcol[0]};
  // (V: the LDS slot a lane fills holds tile row v_row_of_slot(slot))
  const int vrow[2] = {v_row_of_slot(prow[0]), v_row_of_slot(prow[1])};
  const unsigned voffV[2] = {(unsigned)vrow[0] * ldv1 + pcol[0], (unsigned)vrow[0] * ldv2 + pcol[0]};
  const int row3 = v3 ? vrow[1] : prow[1];
  const unsigned voff3[2] = {(unsigned)row3 * (v3 ? ldv1 : ldk1) + pcol[1], (unsigned)row3 * (v3 ? ldv2 : ldk2) + pcol[1]};
  const i32x4 rk1 = make_rsrc(kb0, (unsigned)((int64_t)a.Nk * ldk1)), rv1 = make_rsrc(vb0, (unsigned)((int64_t)a.Nk * ldv1));
  const i32x4 rk2 = make_rsrc(has2 ? a.k2 : kb0, has2 ? (unsigned)((int64_t)a.Nk2 * ldk2) : 0u);
  const i32x4 rv2 = make_rsrc(has2 ? a.v2 : vb0, has2 ? (unsigned)((int64_t)a.Nk2 * ldv2) : 0u);
  constexpr unsigned OOBA = 0x80000000u;  // stays out of range whether or not the (< 2 GiB) scalar offset takes part in the check
  auto dma = [&](const i32x4& r, unsigned voff, unsigned soff, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
  };
  auto issue_tile = [&](int t, int buf) {
    const bool s2 = t >= T0;  // wave-uniform
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;
    const i32x4 rk = s2 ? rk2 : rk1, rv = s2 ? rv2 : rv1;
    const unsigned soffk = (unsigned)kv0 * (s2 ? ldk2 : ldk1), soffv = (unsigned)kv0 * (s2 ? ldv2 : ldv1);
    unsigned oK = s2 ? voffK[1] : voffK[0], oV = s2 ? voffV[1] : voffV[0], o3 = s2 ? voff3[1] : voff3[0];
    if (kv0 + KV_TILE > nk) {  // ragged last tile of a segment: rows past the end read zeros
      if (kv0 + prow[0] >= nk) oK = OOBA;
      if (kv0 + vrow[0] >= nk) oV = OOBA;
      if (kv0 + row3 >= nk) o3 = OOBA;
    }
    const unsigned dK = smem_base + (unsigned)(K_OFF + buf * TILEB), dV = smem_base + (unsigned)(V_OFF + buf * TILEB);
    if (NW == 4) {
      dma(rk, oK, soffk, dK + 1024u * (unsigned)pA);
      dma(rv, oV, soffv, dV + 1024u * (unsigned)pA);
    } else {
      dma(vA ? rv : rk, vA ? oV : oK, vA ? soffv : soffk, (vA ? dV : dK) + 1024u * (unsigned)pA);
    }
    if (has3) dma(v3 ? rv : rk, o3, v3 ? soffv : soffk, (v3 ? dV : dK) + 4096u);
  };
  // wait until only this wave's DMAs of the NEWEST tile may still be in flight
  auto wait_all_but_newest = [&]() {
    if (has3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ - 1) : "memory");
  };

  // ---- register staging (STAGE = 1): thread `tid` moves K chunk tid and V chunk tid; threads < 64 also K chunk 256 + tid,
  // threads 64..127 V chunk 192 + tid.  The LDS image is the same lane-linear row-major tile the DMA writes. ----
  u32x4 sreg[3];
  const int srow0 = tid / 5, srow2 = (256 + (tid & 63)) / 5;
  const unsigned scol0 = (unsigned)((head * D + (tid - srow0 * 5) * 8) * 2), scol2 = (unsigned)((head * D + ((256 + (tid & 63)) - srow2 * 5) * 8) * 2);
  const bool s3 = tid < 128, s3v = tid >= 64;  // has a third chunk / it is a V chunk
  auto stage_loads = [&](int t) {
    const bool s2 = t >= T0;
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;
    const unsigned ldkb = s2 ? ldk2 : ldk1, ldvb = s2 ? ldv2 : ldv1;
    const __amdgpu_buffer_rsrc_t bk = __builtin_amdgcn_make_buffer_rsrc((void*)(s2 ? a.k2 : kb0), 0, (int)((int64_t)nk * ldkb), 0x00020000);
    const __amdgpu_buffer_rsrc_t bv = __builtin_amdgcn_make_buffer_rsrc((void*)(s2 ? a.v2 : vb0), 0, (int)((int64_t)nk * ldvb), 0x00020000);
    const int vr0 = v_row_of_slot(srow0), vr2 = v_row_of_slot(srow2);
    const bool ok0 = kv0 + srow0 < nk, ok2 = kv0 + srow2 < nk, okv0 = kv0 + vr0 < nk, okv2 = kv0 + vr2 < nk;
    sreg[0] = __builtin_amdgcn_raw_buffer_load_b128(bk, ok0 ? (unsigned)srow0 * ldkb + scol0 : OOBA, (unsigned)kv0 * ldkb, 0);
    sreg[1] = __builtin_amdgcn_raw_buffer_load_b128(bv, okv0 ? (unsigned)vr0 * ldvb + scol0 : OOBA, (unsigned)kv0 * ldvb, 0);
    if (s3) {
      if (s3v) sreg[2] = __builtin_amdgcn_raw_buffer_load_b128(bv, okv2 ? (unsigned)vr2 * ldvb + scol2 : OOBA, (unsigned)kv0 * ldvb, 0);
      else sreg[2] = __builtin_amdgcn_raw_buffer_load_b128(bk, ok2 ? (unsigned)srow2 * ldkb + scol2 : OOBA, (unsigned)kv0 * ldkb, 0);
    }
  };
  auto stage_write = [&](int buf) {
    unsigned char* kd = smem + K_OFF + buf * TILEB;
    unsigned char* vd = smem + V_OFF + buf * TILEB;
    *reinterpret_cast<u32x4*>(kd + 16 * tid) = sreg[0];
    *reinterpret_cast<u32x4*>(vd + 16 * tid) = sreg[1];
    if (s3) *reinterpret_cast<u32x4*>((s3v ? vd : kd) + 16 * (256 + (tid & 63))) = sreg[2];
  };

  // per-lane fragment addresses inside ring slot 0 (loop-invariant).  K (A operand, 32x32x16): row 32 u + li, bytes
  // 32 s + 16 h2; lanes h2 = 1 read columns 40..47 of k-step 2 from the constant slot, whose address does not move.
  const unsigned kaddr = (unsigned)(K_OFF + li * ROWB + 16 * h2);
  const unsigned kaddr2_0 = h2 ? (unsigned)CONST_OFF : kaddr + 64;
  const unsigned kaddr2_d = h2 ? 0u : (unsigned)TILEB;
  // V^T (A operand, 16x16x32) by transposed read: lane m = i16 of group g supplies the address of 4 consecutive d of kv row
  // {0, 16, 4, 20}[g] + (m >> 2), d = 16 dt + 4 (m & 3) ..; the hardware hands lane i the column d = 16 dt + i of those
  // 4 kv rows.  For dt = 2 the chunks m & 3 = 2, 3 (d = 40..47) are the constant ones-row / zero slots.
  const int kvb = 16 * (g & 1) + 4 * (g >> 1);
  const unsigned vaddr = smem_base + (unsigned)(V_OFF + v_slot_of_row(kvb + (i16 >> 2)) * ROWB + 8 * (i16 & 3));
  constexpr int HI = MIMO_ATTN40_VPERM ? 16 : 8;   // slot distance of tile rows r and r + 8
  const bool vconst = (i16 & 3) >= 2;
  const unsigned vaddr2 = vconst ? smem_base + (unsigned)(CONST_OFF + 16 + 8 * ((i16 & 3) - 2)) : vaddr + 64;
  const unsigned vstep2 = vconst ? 0u : 1u;

  f32x16 zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  // S^T(x) = K.Q^T(x): 32 kv x 32 q per sub-tile u, three k-steps of 16 (the third carries the reference row)
  auto qk = [&](const uint4 (&kf)[2][3], int x, f32x16 (&st)[2]) {
    if constexpr (ABL == 4) {  // (ablation: no MFMAs — scores from the operands' first registers so that the loads stay live)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[u][r] = -8.f + 1e-30f * (float)(kf[u][r % 3].x ^ qf[x][r % 3].y);
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 3; ++s) st[u] = HT<DT>::mfma32(kf[u][s], qf[x][s], s == 0 ? zero16 : st[u]);
  };
  // exp2 + pack; returns the OR of the packed words (bit 14 of a half set <=> that p >= 2)
  auto exp_pack = [&](const f32x16 (&st)[2], uint32_t (&w)[2][8]) -> uint32_t {
    uint32_t orr = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (ABL == 3) w[u][k] = pack2<DT>(st[u][2 * k] * 0.001f, st[u][2 * k + 1] * 0.001f);  // (ablation: no exponentials)
        else w[u][k] = pack2<DT>(fast_exp2(st[u][2 * k]), fast_exp2(st[u][2 * k + 1]));
        orr |= w[u][k];  // (the compiler pairs these into v_or3_b32)
      }
    return orr;
  };
  // Slow path of one query block (first tile, overshoot, ragged tile): recompute S, mask, move the reference to
  // ceil(running max) + HEAD, rescale O, redo exp + pack.  Exact softmax arithmetic; identical to the fast path's
  // result whenever the fast path is valid.
  auto slow = [&](const uint4 (&kf)[2][3], int x, int t, int kv0, int nk, uint32_t (&w)[2][8]) -> float {
    f32x16 st[2];
    qk(kf, x, st);
    if (kv0 + KV_TILE > nk) {
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
    // st = score - mq; the new reference is ceil(score max) + HEAD, but it never moves down (a lower reference
    // would need no rescale of O yet buys nothing) except on the first tile, where there is no history
    float dlt = ceilf(mt) + HEAD;
    if (t != 0) dlt = fmaxf(dlt, 0.f);
    if (!(dlt > -4000.f)) dlt = 0.f;  // a fully masked tile (mt = -inf) keeps the reference
    dlt = fminf(fmaxf(dlt, -2000.f - mq[x]), 2000.f - mq[x]);
    dlt = HT<DT>::to_f(HT<DT>::from_f(mq[x] + dlt)) - mq[x];  // move by the difference of the STORED halfs
    if (t != 0) {
      const float alpha = fast_exp2(-dlt);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float aj = __shfl(alpha, i16 + 16 * j, 64);  // 32x32 view (q = li) -> 16x16 view (q = 16 j + i16)
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) ot[x][dt][j] *= aj;
      }
    }
    mq[x] += dlt;
    if (h2 == 1) qf[x][2].x = (qf[x][2].x & 0xffff0000u) | (uint32_t)HT<DT>::from_f(-mq[x]);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        w[u][k] = pack2<DT>(fast_exp2(st[u][2 * k] - dlt), fast_exp2(st[u][2 * k + 1] - dlt));
    return dlt;  // how far this lane's reference moved (PIPE: the scores of tile t + 1 were taken against the old one)
  };
  // packed P words of the 32x32 layout -> B operands of the 16x16x32 MFMA.  w[k] (k < 4) holds kv {0..3, 8..11} + 4 h2
  // and w[4 + k] kv {16..19, 24..27} + 4 h2 of query li.  Swapping the odd 16-lane rows of w[k] with the even rows of
  // w[4 + k] leaves in w[k] the fragment of query tile 0 (q = i16) and in w[4 + k] that of query tile 1 (q = 16 + i16);
  // lane group g then holds the kv set {0, 16, 4, 20}[g] + {0..3, 8..11} — the rows the V^T reads above gather.
  auto to_frag = [&](uint32_t (&w)[2][8], uint4 (&pf)[2][2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const auto r2 = __builtin_amdgcn_permlane16_swap(w[u][k], w[u][4 + k], false, false);
        w[u][k] = r2[0];
        w[u][4 + k] = r2[1];
      }
      pf[u][0] = make_uint4(w[u][0], w[u][1], w[u][2], w[u][3]);
      pf[u][1] = make_uint4(w[u][4], w[u][5], w[u][6], w[u][7]);
    }
  };

  [[maybe_unused]] uint4 kf_carried[2][3];  // PIPE = 4: K fragments of the NEXT tile, loaded before the current tile's P.V MFMAs
  // ---- one KV tile; CUR (the ring slot it reads) is a compile-time constant ----
  auto tile = [&](auto cur_c, int t) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int RD = ABL == 1 ? 0 : CUR;   // (ablation: every tile reads slot 0)
    if (t + 2 < T && ABL != 1) {  // slot (CUR + 2) % NB was last read in iteration t-1 (barrier passed)
      if (STAGE == 1) stage_loads(t + 2);
      else issue_tile(t + 2, (CUR + 2) % NB);
    }
    const bool s2 = t >= T0;
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;
    const bool special = (t == 0) | (kv0 + KV_TILE > nk);  // wave-uniform: first / ragged tiles take the slow path
    const unsigned char* kbuf = smem + RD * TILEB;

    uint4 kf_local[2][3];
    uint4 (&kf)[2][3] = PIPE == 4 ? kf_carried : kf_local;   // PIPE = 4: the fragments were fetched under the previous tile's P.V
    if constexpr (PIPE != 4) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        kf[u][0] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB);
        kf[u][1] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB + 32);
        kf[u][2] = *reinterpret_cast<const uint4*>(smem + kaddr2_0 + RD * kaddr2_d + (h2 ? 0 : u * 32 * ROWB));
      }
    }
    f32x16 sa[2], sb[2];
    uint32_t wa[2][8], wb[2][8];
    uint4 pa[2][2], pb[2][2];
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    qk(kf, 0, sa);
    if constexpr (PIPE != 3) qk(kf, 1, sb);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    // V^T fragments of the whole tile: 12 transposed reads issued now, waited for after the softmax (the compiler does
    // not track asm loads: the wait statement below names every destination)
    uint2 vlo[3][2], vhi[3][2];
    {
      const unsigned va2 = vaddr2 + (unsigned)(RD * TILEB) * vstep2;
#define MIMO_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        MIMO_TR(vlo[0][u], vaddr, RD * TILEB + 32 * u * ROWB);
        MIMO_TR(vhi[0][u], vaddr, RD * TILEB + (32 * u + HI) * ROWB);
        MIMO_TR(vlo[1][u], vaddr, RD * TILEB + 32 * u * ROWB + 32);
        MIMO_TR(vhi[1][u], vaddr, RD * TILEB + (32 * u + HI) * ROWB + 32);
      }
      // dt = 2: lanes of the constant chunks must not move with u / hi: their offsets are applied through vstep2
      {
        const unsigned a00 = va2, a01 = va2 + (HI * ROWB) * vstep2, a10 = va2 + (32 * ROWB) * vstep2, a11 = va2 + ((32 + HI) * ROWB) * vstep2;
        MIMO_TR(vlo[2][0], a00, 0);
        MIMO_TR(vhi[2][0], a01, 0);
        MIMO_TR(vlo[2][1], a10, 0);
        MIMO_TR(vhi[2][1], a11, 0);
      }
#undef MIMO_TR
    }
    if constexpr (PIPE == 3) {
      // Staggered query blocks (round 5, no extra registers, same tile granularity): the wave issues in order, so the scores of
      // block B are multiplied while block A is exponentiated, and P.V of block A runs under the exponentials of block B:
      //   Q.K^T(A) | Q.K^T(B) || exp(A) | P.V(A) || exp(B) | P.V(B)
      __builtin_amdgcn_sched_barrier(0);
      qk(kf, 1, sb);
      const uint32_t ora = exp_pack(sa, wa);
#define MIMO_SGB(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
      MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(10); MIMO_SGB(10);
#undef MIMO_SGB
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_amdgcn_ballot_w64(special | ((ora & 0x40004000u) != 0u)) != 0ull) slow(kf, 0, t, kv0, nk, wa);
      to_frag(wa, pa);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vlo[0][0]), "+v"(vlo[0][1]), "+v"(vlo[1][0]), "+v"(vlo[1][1]), "+v"(vlo[2][0]), "+v"(vlo[2][1]),
                     "+v"(vhi[0][0]), "+v"(vhi[0][1]), "+v"(vhi[1][0]), "+v"(vhi[1][1]), "+v"(vhi[2][0]), "+v"(vhi[2][1])
                   :: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 vf = make_uint4(vlo[dt][u].x, vlo[dt][u].y, vhi[dt][u].x, vhi[dt][u].y);
#pragma unroll
          for (int j = 0; j < 2; ++j) ot[0][dt][j] = HT<DT>::mfma16(vf, pa[u][j], ot[0][dt][j]);
        }
      const uint32_t orb = exp_pack(sb, wb);
#define MIMO_SGB(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x002, n, 1)
      MIMO_SGB(6); MIMO_SGB(6); MIMO_SGB(6); MIMO_SGB(6); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5); MIMO_SGB(5);
#undef MIMO_SGB
      __builtin_amdgcn_sched_barrier(0);
      if (__builtin_amdgcn_ballot_w64(special | ((orb & 0x40004000u) != 0u)) != 0ull) slow(kf, 1, t, kv0, nk, wb);
      to_frag(wb, pb);
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 vf = make_uint4(vlo[dt][u].x, vlo[dt][u].y, vhi[dt][u].x, vhi[dt][u].y);
#pragma unroll
          for (int j = 0; j < 2; ++j) ot[1][dt][j] = HT<DT>::mfma16(vf, pb[u][j], ot[1][dt][j]);
        }
    } else {
    const uint32_t ora = exp_pack(sa, wa);
    const uint32_t orb = exp_pack(sb, wb);
    if (__builtin_amdgcn_ballot_w64(special | ((ora & 0x40004000u) != 0u)) != 0ull) slow(kf, 0, t, kv0, nk, wa);
    if (__builtin_amdgcn_ballot_w64(special | ((orb & 0x40004000u) != 0u)) != 0ull) slow(kf, 1, t, kv0, nk, wb);
    to_frag(wa, pa);
    to_frag(wb, pb);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vlo[0][0]), "+v"(vlo[0][1]), "+v"(vlo[1][0]), "+v"(vlo[1][1]), "+v"(vlo[2][0]), "+v"(vlo[2][1]),
                   "+v"(vhi[0][0]), "+v"(vhi[0][1]), "+v"(vhi[1][0]), "+v"(vhi[1][1]), "+v"(vhi[2][0]), "+v"(vhi[2][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);  // no MFMA may be hoisted above the wait (cdna_hip_programming.md rule 18)
    if constexpr (PIPE == 4) {
      // Every LDS read of slot CUR is complete (K fragments before Q.K^T, V^T fragments just now): the tile's barrier moves
      // HERE, in front of P.V, and the K fragments of tile t + 1 are requested right behind it — their LDS latency and the
      // barrier skew hide under the 24 P.V MFMAs instead of standing in front of the next tile's first MFMA.
      if (t + 1 < T) {
        if (t + 2 < T) wait_all_but_newest();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        constexpr int NX = (CUR + 1) % NB;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          kf_carried[u][0] = *reinterpret_cast<const uint4*>(smem + NX * TILEB + kaddr + u * 32 * ROWB);
          kf_carried[u][1] = *reinterpret_cast<const uint4*>(smem + NX * TILEB + kaddr + u * 32 * ROWB + 32);
          kf_carried[u][2] = *reinterpret_cast<const uint4*>(smem + kaddr2_0 + NX * kaddr2_d + (h2 ? 0 : u * 32 * ROWB));
        }
      }
    }
    // ---- O^T += V^T.P^T: d tiles {0..15, 16..31, 32..47}, two k-steps of 32 kv; every V^T fragment feeds 4 query tiles ----
    if (PRIO != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint4 vf = make_uint4(vlo[dt][u].x, vlo[dt][u].y, vhi[dt][u].x, vhi[dt][u].y);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ABL == 4) {  // (ablation: no MFMAs)
            ot[0][dt][j][0] += 1e-30f * (float)(vf.x ^ pa[u][j].x);
            ot[1][dt][j][0] += 1e-30f * (float)(vf.y ^ pb[u][j].y);
          } else {
            ot[0][dt][j] = HT<DT>::mfma16(vf, pa[u][j], ot[0][dt][j]);
            ot[1][dt][j] = HT<DT>::mfma16(vf, pb[u][j], ot[1][dt][j]);
          }
        }
      }
    if (PRIO != 0) __builtin_amdgcn_s_setprio(0);
    }  // PIPE != 3
    if (PIPE != 4 && t + 1 < T) {
      if (STAGE == 1) {
        if (t + 2 < T && ABL != 1) stage_write((CUR + 2) % NB);  // tile t+2: loaded at the top, readable after two barriers
      } else {
        // tile t+1 (issued one iteration ago) has to be in LDS; tile t+2 (issued above) may stay in flight
        if (ABL == 2) {  // (ablation: DMAs issued, never waited for — the tiles read may be stale)
        } else if (t + 2 < T && ABL != 1) wait_all_but_newest();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();  // publishes the staged tiles; every wave is done reading slot CUR
    }
  };

  // ---- PIPE = 1: one KV tile of the software-pipelined form.  S[P] holds the scores of tile t (computed one tile earlier),
  // S[P ^ 1] receives those of tile t + 1 (MORE = false: t is the last tile).  CUR = t % 4 is the ring slot of tile t. ----
  [[maybe_unused]] f32x16 S[2][2][2];  // [parity][query block x][kv sub-tile u]
  auto load_kf = [&](int slot, uint4 (&kf)[2][3]) {
    const unsigned char* kbuf = smem + slot * TILEB;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kf[u][0] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB);
      kf[u][1] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB + 32);
      kf[u][2] = *reinterpret_cast<const uint4*>(smem + kaddr2_0 + slot * kaddr2_d + (h2 ? 0 : u * 32 * ROWB));
    }
  };
  // exp2 + pack of one 32 x 32 score sub-tile; returns the OR of the packed words
  auto exp_pack_u = [&](const f32x16& st, uint32_t (&w)[8]) -> uint32_t {
    uint32_t orr = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      w[k] = pack2<DT>(fast_exp2(st[2 * k]), fast_exp2(st[2 * k + 1]));
      orr |= w[k];
    }
    return orr;
  };
  auto tile_p = [&](auto cur_c, auto par_c, auto more_c, int t) {
    constexpr int CUR = decltype(cur_c)::value, P = decltype(par_c)::value;
    constexpr bool MORE = decltype(more_c)::value != 0;
    if (t + 3 < T) issue_tile(t + 3, (CUR + 3) % NB);  // slot (CUR + 3) % 4 held V of tile t - 1: last read before the previous barrier
    const bool s2 = t >= T0;
    const int nk = s2 ? a.Nk2 : a.Nk;
    const int kv0 = (s2 ? t - T0 : t) * KV_TILE;
    const bool special = (t == 0) | (kv0 + KV_TILE > nk);  // wave-uniform: first / ragged tiles take the slow path
    // K fragments of tile t + 1, one kv half (u) at a time: 12 registers live instead of 24
    uint4 kf[3];
    auto load_ku = [&](int u) {
      const unsigned char* kbuf = smem + ((CUR + 1) % NB) * TILEB;
      kf[0] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB);
      kf[1] = *reinterpret_cast<const uint4*>(kbuf + kaddr + u * 32 * ROWB + 32);
      kf[2] = *reinterpret_cast<const uint4*>(smem + kaddr2_0 + ((CUR + 1) % NB) * kaddr2_d + (h2 ? 0 : u * 32 * ROWB));
    };
    // Q.K^T of tile t + 1 for kv half u: two independent accumulator chains (query blocks A, B), interleaved
    auto qk_u = [&](int u) {
#pragma unroll
      for (int sk = 0; sk < 3; ++sk)
#pragma unroll
        for (int x = 0; x < 2; ++x)
          S[P ^ 1][x][u] = HT<DT>::mfma32(kf[sk], qf[x][sk], sk == 0 ? zero16 : S[P ^ 1][x][u]);
    };
    uint32_t wa[2][8], wb[2][8];
    uint4 pa[2][2], pb[2][2];
    uint2 vlo[3][2], vhi[3][2];
    // ---- phase B1: Q.K^T of tile t + 1, kv half 0 (6 MFMAs of 32 matrix-pipe cycles) under the exponentials of query block A
    // (64 VALU instructions) ----
    if constexpr (MORE) {
      load_ku(0);
      qk_u(0);
    }
    uint32_t ora = exp_pack_u(S[P][0][0], wa[0]);
    ora |= exp_pack_u(S[P][0][1], wa[1]);
    if constexpr (MORE) {
      // (one MFMA, then its share of the exp / pack / or stream: 64 VALU instructions over 6 MFMAs)
#define MIMO_SGB(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
      MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(10); MIMO_SGB(10);
#undef MIMO_SGB
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B2: kv half 1 under the exponentials of query block B; the V^T fragments of tile t are requested here (12
    // transposed reads, waited for before P.V; the compiler does not track asm loads: the wait statement names them) ----
    if constexpr (MORE) {
      load_ku(1);
      qk_u(1);
    }
    {
      const unsigned va2 = vaddr2 + (unsigned)(CUR * TILEB) * vstep2;
#define MIMO_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        MIMO_TR(vlo[0][u], vaddr, CUR * TILEB + 32 * u * ROWB);
        MIMO_TR(vhi[0][u], vaddr, CUR * TILEB + (32 * u + HI) * ROWB);
        MIMO_TR(vlo[1][u], vaddr, CUR * TILEB + 32 * u * ROWB + 32);
        MIMO_TR(vhi[1][u], vaddr, CUR * TILEB + (32 * u + HI) * ROWB + 32);
      }
      {
        const unsigned a00 = va2, a01 = va2 + (HI * ROWB) * vstep2, a10 = va2 + (32 * ROWB) * vstep2, a11 = va2 + ((32 + HI) * ROWB) * vstep2;
        MIMO_TR(vlo[2][0], a00, 0);
        MIMO_TR(vhi[2][0], a01, 0);
        MIMO_TR(vlo[2][1], a10, 0);
        MIMO_TR(vhi[2][1], a11, 0);
      }
#undef MIMO_TR
    }
    uint32_t orb = exp_pack_u(S[P][1][0], wb[0]);
    orb |= exp_pack_u(S[P][1][1], wb[1]);
    if constexpr (MORE) {
      // (one MFMA, then its share of the exp / pack / or stream: 64 VALU instructions over 6 MFMAs)
#define MIMO_SGB(n) __builtin_amdgcn_sched_group_barrier(0x008, 1, 1); __builtin_amdgcn_sched_group_barrier(0x002, n, 1)
      MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(11); MIMO_SGB(10); MIMO_SGB(10);
#undef MIMO_SGB
    }
    __builtin_amdgcn_sched_barrier(0);
    float dlt_a = 0.f, dlt_b = 0.f;
    if (__builtin_amdgcn_ballot_w64(special | ((ora & 0x40004000u) != 0u)) != 0ull) {
      uint4 kc[2][3];
      load_kf(CUR, kc);
      dlt_a = slow(kc, 0, t, kv0, nk, wa);
    }
    if (__builtin_amdgcn_ballot_w64(special | ((orb & 0x40004000u) != 0u)) != 0ull) {
      uint4 kc[2][3];
      load_kf(CUR, kc);
      dlt_b = slow(kc, 1, t, kv0, nk, wb);
    }
    to_frag(wa, pa);
    to_frag(wb, pb);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vlo[0][0]), "+v"(vlo[0][1]), "+v"(vlo[1][0]), "+v"(vlo[1][1]), "+v"(vlo[2][0]), "+v"(vlo[2][1]),
                   "+v"(vhi[0][0]), "+v"(vhi[0][1]), "+v"(vhi[1][0]), "+v"(vhi[1][1]), "+v"(vhi[2][0]), "+v"(vhi[2][1])
                 :: "memory");
    __builtin_amdgcn_sched_barrier(0);  // no MFMA may be hoisted above the wait
    // ---- phase E: O^T += V^T.P^T, 24 MFMAs of 16 matrix-pipe cycles; every V^T fragment feeds 4 query tiles ----
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint4 vf = make_uint4(vlo[dt][u].x, vlo[dt][u].y, vhi[dt][u].x, vhi[dt][u].y);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          ot[0][dt][j] = HT<DT>::mfma16(vf, pa[u][j], ot[0][dt][j]);
          ot[1][dt][j] = HT<DT>::mfma16(vf, pb[u][j], ot[1][dt][j]);
        }
      }
    if constexpr (MORE) {
      // the scores of tile t + 1 were taken against the reference a slow path may just have moved (rare: wave-uniform skip)
      if (__builtin_amdgcn_ballot_w64((dlt_a != 0.f) | (dlt_b != 0.f)) != 0ull) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            S[P ^ 1][0][u][r] -= dlt_a;
            S[P ^ 1][1][u][r] -= dlt_b;
          }
      }
      // tile t + 2 (issued two iterations ago) has to be in LDS for the next iteration's Q.K^T; tile t + 3 may stay in flight
      if (t + 3 < T) wait_all_but_newest();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // publishes the staged tiles; every wave is done reading K of slot CUR + 1 and V of slot CUR
    }
  };

  if constexpr (PIPE == 1 || PIPE == 2) {
    issue_tile(0, 0);
    if (T > 1) issue_tile(1, 1);
    if (T > 2) issue_tile(2, 2);
    if (T > 2) wait_all_but_newest();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tiles 0 and 1 and the constant slots are visible
    {
      uint4 kf[2][3];
      load_kf(0, kf);
      qk(kf, 0, S[0][0]);
      qk(kf, 1, S[0][1]);
    }
    int t = 0;
    for (; t + 4 < T; t += 4) {
      tile_p(IC2<0>{}, IC2<0>{}, IC2<1>{}, t);
      tile_p(IC2<1>{}, IC2<1>{}, IC2<1>{}, t + 1);
      tile_p(IC2<2>{}, IC2<0>{}, IC2<1>{}, t + 2);
      tile_p(IC2<3>{}, IC2<1>{}, IC2<1>{}, t + 3);
    }
    const int rem = T - t;  // 1..4 tiles left (t is a multiple of 4: slot 0, parity 0)
    if (rem == 1) {
      tile_p(IC2<0>{}, IC2<0>{}, IC2<0>{}, t);
    } else if (rem == 2) {
      tile_p(IC2<0>{}, IC2<0>{}, IC2<1>{}, t);
      tile_p(IC2<1>{}, IC2<1>{}, IC2<0>{}, t + 1);
    } else if (rem == 3) {
      tile_p(IC2<0>{}, IC2<0>{}, IC2<1>{}, t);
      tile_p(IC2<1>{}, IC2<1>{}, IC2<1>{}, t + 1);
      tile_p(IC2<2>{}, IC2<0>{}, IC2<0>{}, t + 2);
    } else {
      tile_p(IC2<0>{}, IC2<0>{}, IC2<1>{}, t);
      tile_p(IC2<1>{}, IC2<1>{}, IC2<1>{}, t + 1);
      tile_p(IC2<2>{}, IC2<0>{}, IC2<1>{}, t + 2);
      tile_p(IC2<3>{}, IC2<1>{}, IC2<0>{}, t + 3);
    }
  } else {
  // ---- pipeline: two tiles in flight ahead of the one being consumed; ONE barrier per tile ----
  if (STAGE == 1) {
    stage_loads(0);
    stage_write(0);
    if (T > 1) {
      stage_loads(1);
      stage_write(1);
    }
  } else {
    issue_tile(0, 0);
    if (T > 1) issue_tile(1, 1);
    if (T > 1) wait_all_but_newest();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();  // tiles 0 (and 1) and the constant slots are visible
  if constexpr (PIPE == 4) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kf_carried[u][0] = *reinterpret_cast<const uint4*>(smem + kaddr + u * 32 * ROWB);
      kf_carried[u][1] = *reinterpret_cast<const uint4*>(smem + kaddr + u * 32 * ROWB + 32);
      kf_carried[u][2] = *reinterpret_cast<const uint4*>(smem + kaddr2_0 + (h2 ? 0 : u * 32 * ROWB));
    }
  }

  for (int t = 0; t < T; t += 3) {
    tile(IC2<0>{}, t);
    if (t + 1 < T) tile(IC2<1>{}, t + 1);
    if (t + 2 < T) tile(IC2<2>{}, t + 2);
  }
  }  // PIPE == 0

  // ---- epilogue: denominators from the ones-row (d = 40: tile 2, row 8 -> lane group 2, register 0) ----
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float den = __shfl(ot[x][2][j][0], 32 + i16, 64);
      const float inv = den > 0.f ? 1.f / den : 0.f;
      const int qr = q0 + 32 * x + 16 * j + i16;
      if (qr < a.Nq) {
        uint16_t* op = a.out + ((int64_t)b * a.Nq + qr) * a.ldo + head * D + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          if (dt == 2 && g >= 2) continue;  // rows 40..47 are not output columns
          uint2 o;
          o.x = pack2<DT>(ot[x][dt][j][0] * inv, ot[x][dt][j][1] * inv);
          o.y = pack2<DT>(ot[x][dt][j][2] * inv, ot[x][dt][j][3] * inv);
          *reinterpret_cast<uint2*>(op + 16 * dt) = o;
        }
      }
    }
}

// ------------------------------------------------------------------------------------
// attn512_kernel: ONE head of d = 512 — the mid-block attention of the VAE (diffusers UNetMidBlock2D, 1-head
// Attention(residual_connection) over the H*W tokens of a latent image: 4096 tokens at 512x512, 9216 at 768x768,
// 9604 at 784x784).  Flash form: the N x N score matrix (340 MB per image at 768x768 in the GEMM + softmax + GEMM
// form this replaces) never exists.  A lane cannot own a 512-wide output row (256 accumulator registers), so the head
// dimension is split over the 4 waves of a block: wave w owns channels [128 w, 128 w + 128) of Q, K, V and O for the
// block's 32 queries.  Per 64-key tile every wave
//   1. computes the PARTIAL S^T = K[:, slice].Q[:, slice]^T of its slice (K fragments straight from global memory: a
//      fragment is 16 contiguous bytes per lane, all query blocks of an image re-read the same K from L2),
//   2. exchanges it through LDS: the four partials are summed in one fixed order by every wave (all waves hold the
//      SAME full scores, hence the same running max / sum: no second exchange),
//   3. runs the online softmax redundantly and multiplies P^T into its own 128 x 32 slice of O^T, V^T fragments coming
//      out of a wave-private row-major V patch by ds_read_b64_tr_b16 (as in temporal_attn2_kernel).
// ------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256, 1) void attn512_kernel(const AttnArgs a) {
  constexpr int D = 512, DW = 128, KS = DW / 16, OT = DW / 32;
  constexpr int VPB = DW * 2;                // V patch row pitch in bytes
  constexpr int VB = KV_TILE * VPB;          // 16 KB per wave
  constexpr int SB = 8 * 64 * 16;            // one wave's partial S^T: 8 x f32x4 per lane = 8 KB
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * VB + 4 * SB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, li = lane & 31, g = lane >> 4, i16 = lane & 15;
  // (query blocks of one image on one XCD: see attn_kernel)
  const unsigned Lr_ = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const int b = (int)(Lr_ / gridDim.x);
  const int q0 = (int)(Lr_ % gridDim.x) * 32;
  const int dw0 = wave * DW;
  unsigned char* const vpatch = smem + wave * VB;
  f32x4* const sred = reinterpret_cast<f32x4*>(smem + 4 * VB);

  const uint16_t* qb = a.q + (int64_t)b * a.Nq * a.ldq;
  const uint16_t* kb = a.k + (int64_t)b * a.Nk * a.ldk;
  const uint16_t* vb = a.v + (int64_t)b * a.Nk * a.ldv;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (int)((int64_t)a.Nq * a.ldq * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)((int64_t)a.Nk * a.ldk * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, (int)((int64_t)a.Nk * a.ldv * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // Q^T fragments of this wave's channel slice (B operand): lane (h2, q = li) holds Q[q][dw0 + 16 s + 8 h2 .. + 8]
  uint4 qf[KS];
  {
    const int qr = q0 + li;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
          rq, qr < a.Nq ? (unsigned)(((int64_t)qr * a.ldq + dw0 + 16 * s + 8 * h2) * 2) : OOB, 0, 0);
      qf[s] = make_uint4(v.x, v.y, v.z, v.w);
    }
  }
  f32x16 ot[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c = a.scale_log2;
  const int T = (a.Nk + KV_TILE - 1) / KV_TILE;
  const unsigned ldkb = (unsigned)(a.ldk * 2), ldvb = (unsigned)(a.ldv * 2);
  const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)vpatch +
                         (unsigned)((4 * h2 + (i16 >> 2)) * VPB + (16 * (g & 1) + 4 * (i16 & 3)) * 2);

  for (int t = 0; t < T; ++t) {
    const int kv0 = t * KV_TILE;
    // ---- global loads of the tile: K fragments (A operand: lane (h2, kv = li)) and V chunks (row 4 j + (lane >> 4),
    // 16-byte chunk lane & 15 of the 128-channel slice); rows past the end read zeros ----
    uint4 kf[2][KS];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kr = kv0 + 32 * u + li;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
            rk, kr < a.Nk ? (unsigned)kr * ldkb + (unsigned)((dw0 + 16 * s + 8 * h2) * 2) : OOB, 0, 0);
        kf[u][s] = make_uint4(v.x, v.y, v.z, v.w);
      }
    }
    u32x4 vreg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int vr = kv0 + 4 * j + (lane >> 4);
      vreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rv, vr < a.Nk ? (unsigned)vr * ldvb + (unsigned)((dw0 + 8 * (lane & 15)) * 2) : OOB, 0, 0);
    }
    // ---- partial S^T over this wave's 128 channels ----
    f32x16 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) st[u] = HT<DT>::mfma32(kf[u][s], qf[s], st[u]);
    }
    // ---- exchange: every wave sums the four partials in the order 0, 1, 2, 3 ----
    if (t > 0) __syncthreads();  // the previous tile's partials have been read by every wave
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        sred[(wave * 8 + u * 4 + i) * 64 + lane] = (f32x4){st[u][4 * i], st[u][4 * i + 1], st[u][4 * i + 2], st[u][4 * i + 3]};
    // (the V rows are parked in the wave-private patch while the partials travel; the previous tile's transposed reads
    // were waited for, and DS operations of a wave execute in order)
#pragma unroll
    for (int j = 0; j < 16; ++j)
      *reinterpret_cast<uint4*>(vpatch + (4 * j + (lane >> 4)) * VPB + (lane & 15) * 16) = make_uint4(vreg[j].x, vreg[j].y, vreg[j].z, vreg[j].w);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 sum = sred[(0 * 8 + u * 4 + i) * 64 + lane];
        sum += sred[(1 * 8 + u * 4 + i) * 64 + lane];
        sum += sred[(2 * 8 + u * 4 + i) * 64 + lane];
        sum += sred[(3 * 8 + u * 4 + i) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) st[u][4 * i + r] = sum[r];
      }
    // ---- online softmax (raw-score running max, scale folded into the exp2 argument), as attn_kernel ----
    if (kv0 + KV_TILE > a.Nk) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (kv >= a.Nk) st[u][r] = -INFINITY;
        }
    }
    float mt = st[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[u][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0ull) {
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
        ps += pv;
      }
    ps += __shfl_xor(ps, 32, 64);
    l_run += ps;
    uint4 pf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = u >> 1, hh = u & 1;
      pf[u].x = pack2<DT>(st[tt][8 * hh + 0], st[tt][8 * hh + 1]);
      pf[u].y = pack2<DT>(st[tt][8 * hh + 2], st[tt][8 * hh + 3]);
      pf[u].z = pack2<DT>(st[tt][8 * hh + 4], st[tt][8 * hh + 5]);
      pf[u].w = pack2<DT>(st[tt][8 * hh + 6], st[tt][8 * hh + 7]);
    }
    // ---- O^T[slice] += V^T.P^T: k-slot j of fragment u is key 16 u + 4 h2 + {0,1,2,3,8,9,10,11}[j] ----
#pragma unroll
    for (int dt = 0; dt < OT; ++dt) {
      uint2 tl[4], th[4];
      const unsigned va = vbase + (unsigned)(32 * dt * 2);
#define MIMO_TR512(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "i"(off) : "memory")
      MIMO_TR512(tl[0], 0 * VPB);  MIMO_TR512(th[0], 8 * VPB);
      MIMO_TR512(tl[1], 16 * VPB); MIMO_TR512(th[1], 24 * VPB);
      MIMO_TR512(tl[2], 32 * VPB); MIMO_TR512(th[2], 40 * VPB);
      MIMO_TR512(tl[3], 48 * VPB); MIMO_TR512(th[3], 56 * VPB);
#undef MIMO_TR512
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(th[0]), "+v"(th[1]), "+v"(th[2]), "+v"(th[3])::"memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) ot[dt] = HT<DT>::mfma32(make_uint4(tl[u].x, tl[u].y, th[u].x, th[u].y), pf[u], ot[dt]);
    }
  }
  // ---- epilogue: lane (h2, q) holds O^T[d = dw0 + 32 dt + (r & 3) + 8 (r >> 2) + 4 h2][q] ----
  const int qr = q0 + li;
  if (qr < a.Nq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    uint16_t* op = a.out + ((int64_t)b * a.Nq + qr) * a.ldo + dw0;
#pragma unroll
    for (int dt = 0; dt < OT; ++dt)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int d = 32 * dt + 8 * cc + 4 * h2;
        uint2 o;
        o.x = pack2<DT>(ot[dt][4 * cc + 0] * inv, ot[dt][4 * cc + 1] * inv);
        o.y = pack2<DT>(ot[dt][4 * cc + 2] * inv, ot[dt][4 * cc + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = o;
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
// Temporal attention, second form (the one mimo_temporal_attention launches when the row strides allow 16-byte
// accesses): same wave <-> (batch, pixel, head) mapping and the same Q.K^T / softmax arithmetic as temporal_attn_kernel,
// but every global access is a 16-byte lane access.  The first form gathered V^T with 2-byte loads (32 load instructions
// per lane at d = 40, each lane its own request) and stored O^T as 8-byte pieces: rocprofv3 showed its waves 79 % of
// the time issue-stalled at twice its byte roofline.  Here V rows are fetched as whole 16-byte chunks (2 instructions per
// lane at d = 40), parked row-major in a wave-private LDS patch and read back TRANSPOSED by ds_read_b64_tr_b16 (the
// hardware hands lane i column i of a [4 frames][16 channels] block: exactly the V^T fragment of the 32x32x16 MFMA);
// O goes through a second patch and leaves as 16-byte stores of whole (frame, head) rows.
// The patches are wave-private: DS operations of one wave execute in order, no barrier is needed.
// ------------------------------------------------------------------------------------
template <int DT, int D>
__global__ __launch_bounds__(256, 2) void temporal_attn2_kernel(const TAttnArgs a) {
  constexpr int KS = (D + 15) / 16;
  constexpr int OT = (D + 31) / 32;
  constexpr int DC = D / 8;                 // 16-byte chunks per (frame, head) row
  constexpr int VP = 32 * OT;               // V patch row pitch (halfs): the padded channel count
  constexpr int VB = 32 * VP * 2;           // bytes: 32 frame rows (rows >= F stay zero)
  constexpr int OB = 32 * D * 2;            // bytes: O patch, dense [frame][D]
  constexpr int NV = (32 * DC + 63) / 64;   // 16-byte chunks per lane that cover 32 rows
  static_assert(D % 8 == 0, "rows are whole 16-byte chunks");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * (VB + OB)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, li = lane & 31, g = lane >> 4, i16 = lane & 15;
  unsigned char* const vpatch = smem + wave * (VB + OB);
  unsigned char* const opatch = vpatch + VB;
  const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nunits = (int64_t)a.b * a.HW * a.heads;
  if (unit >= nunits) return;
  const int head = (int)(unit % a.heads);
  const int64_t bp = unit / a.heads;
  const int64_t pix = bp % a.HW;
  const int bi = (int)(bp / a.HW);
  const int F = a.F;
  const int64_t row0 = (int64_t)bi * F * a.HW + pix;  // token row of frame f: row0 + f * HW

  const int64_t tot_rows = (int64_t)a.b * F * a.HW;
  const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)a.q, 0, (int)(tot_rows * a.ldq * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)a.k, 0, (int)(tot_rows * a.ldk * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)a.v, 0, (int)(tot_rows * a.ldv * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- every global load of the unit is issued up front: V chunks first (they take the longest way) ----
  u32x4 vreg[NV];
  int vrow[NV], vcc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = lane + 64 * j;
    vrow[j] = c / DC;
    vcc[j] = c - vrow[j] * DC;
    const bool ok = vrow[j] < F;
    vreg[j] = __builtin_amdgcn_raw_buffer_load_b128(
        rv, ok ? (unsigned)(((row0 + (int64_t)vrow[j] * a.HW) * a.ldv + head * D + vcc[j] * 8) * 2) : OOB, 0, 0);
  }
  u32x4 kv4[KS], qv4[KS];
  {
    const bool fok = li < F;
    const int64_t row = row0 + (int64_t)li * a.HW;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int kk = 16 * s + 8 * h2;
      const bool ok = fok & (kk < D);
      kv4[s] = __builtin_amdgcn_raw_buffer_load_b128(rk, ok ? (unsigned)((row * a.ldk + head * D + kk) * 2) : OOB, 0, 0);
      qv4[s] = __builtin_amdgcn_raw_buffer_load_b128(rq, ok ? (unsigned)((row * a.ldq + head * D + kk) * 2) : OOB, 0, 0);
    }
  }
  // zero the V patch (rows >= F and channels >= D are never written: they must read as zeros, p = 0 times garbage
  // could be NaN), then park the V rows
  for (int i = lane; i < VB / 16; i += 64) reinterpret_cast<uint4*>(vpatch)[i] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (vrow[j] < F)
      *reinterpret_cast<uint4*>(vpatch + (vrow[j] * VP + vcc[j] * 8) * 2) = make_uint4(vreg[j].x, vreg[j].y, vreg[j].z, vreg[j].w);

  // ---- S^T = K.Q^T (32 frames x 32 frames), softmax over the key frames: identical to temporal_attn_kernel ----
  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
    st = HT<DT>::mfma32(make_uint4(kv4[s].x, kv4[s].y, kv4[s].z, kv4[s].w), make_uint4(qv4[s].x, qv4[s].y, qv4[s].z, qv4[s].w), st);
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
  // ---- O^T += V^T.P^T.  k-slot j of fragment u is frame 16 u + 4 h2 + {0,1,2,3,8,9,10,11}[j] (the accumulator layout of
  // S^T).  Transposed read: inside a 16-lane group lane m supplies the address of 4 consecutive channels of frame row
  // (m >> 2), channel chunk (m & 3); lane i receives channel i of those 4 rows.  Group g covers channels 16 (g & 1) ..
  // of the 32-channel tile and frames 4 (g >> 1) .. ----
  const unsigned vbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)vpatch +
                         (unsigned)(((4 * h2 + (i16 >> 2)) * VP + 16 * (g & 1) + 4 * (i16 & 3)) * 2);
#pragma unroll
  for (int dt = 0; dt < OT; ++dt) {
    uint2 t00, t01, t10, t11;
    const unsigned va = vbase + (unsigned)(32 * dt * 2);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(t00) : "v"(va) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t01) : "v"(va), "i"(8 * VP * 2) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t10) : "v"(va), "i"(16 * VP * 2) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t11) : "v"(va), "i"(24 * VP * 2) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t00), "+v"(t01), "+v"(t10), "+v"(t11)::"memory");
    __builtin_amdgcn_sched_barrier(0);  // no MFMA may be hoisted above the wait
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    o = HT<DT>::mfma32(make_uint4(t00.x, t00.y, t01.x, t01.y), pf[0], o);
    o = HT<DT>::mfma32(make_uint4(t10.x, t10.y, t11.x, t11.y), pf[1], o);
    // lane (h2, q = li): o[r] = O^T[d = 32 dt + (r & 3) + 8 (r >> 2) + 4 h2][q] -> O patch row q
    if (li < F) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = 32 * dt + 8 * c + 4 * h2;
        if (dd < D) {
          uint2 w;
          w.x = pack2<DT>(o[4 * c + 0] * inv, o[4 * c + 1] * inv);
          w.y = pack2<DT>(o[4 * c + 2] * inv, o[4 * c + 3] * inv);
          *reinterpret_cast<uint2*>(opatch + (li * D + dd) * 2) = w;
        }
      }
    }
  }
  // ---- whole 16-byte chunks of the (frame, head) rows leave the patch ----
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (vrow[j] < F) {
      const uint4 w = *reinterpret_cast<const uint4*>(opatch + (lane + 64 * j) * 16);
      *reinterpret_cast<uint4*>(a.out + (row0 + (int64_t)vrow[j] * a.HW) * a.ldo + head * D + vcc[j] * 8) = w;
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

static inline bool attn40_legacy() { return tune_env("MIMO_ATTN40_LEGACY", 0) != 0; }
template <int DT>
static inline void attn40_launch(const AttnArgs& a, hipStream_t st) {
  // 4-wave blocks (256 queries) by default: 8-wave blocks halve the DMA work per FLOP but measured 12 % slower
  // (one block per CU: nothing overlaps its per-tile barrier; profiles/r2_attn40_ab.txt)
  const int nw = tune_env("MIMO_ATTN40_NW", 4);
  const dim3 grid((unsigned)(((a.Nq + 64 * nw - 1) / (64 * nw)) * a.heads * a.B));
#ifdef MIMO_TUNE
  if (tune_env("MIMO_ATTN40_ABLATE", 0) == 1) {
    if (nw == 8) hipLaunchKernelGGL((attn40_kernel<DT, 8, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((attn40_kernel<DT, 4, 1>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_ABLATE", 0) == 2) {
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 2>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_ABLATE", 0) == 3) {  // no exponentials (results wrong): what the transcendental stream costs
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 3>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_ABLATE", 0) == 4) {  // no MFMAs (results wrong): what the matrix work costs
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 4>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_STAGE", 0) == 1 && nw == 4) {
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 1>), grid, dim3(256), 0, st, a);
    return;
  }
  if (nw == 8) {
    hipLaunchKernelGGL((attn40_kernel<DT, 8, 0>), grid, dim3(512), 0, st, a);
    return;
  }
#endif
#ifdef MIMO_TUNE
  // round-5 experiment (profiles/r5_attn40_pipe_ab.txt): the software-pipelined tile.  PIPE = 2 (one wave per SIMD, 512
  // registers) is correct and exactly as fast as the shipped form; PIPE = 1 (two waves per SIMD) does not fit 256 registers —
  // its scratch traffic breaks the counted vmcnt waits (wrong results) — and is not instantiated.
  if (tune_env("MIMO_ATTN40_PIPE", 0) == 2) {
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 0, 2>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_PIPE", 0) == 4) {  // K fragments of the next tile fetched under P.V (barrier moved in front of P.V)
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 0, 4>), grid, dim3(256), 0, st, a);
    return;
  }
  if (tune_env("MIMO_ATTN40_PIPE", 0) == 3) {  // staggered query blocks inside the tile (two waves per SIMD, no extra registers)
    hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 0, 3>), grid, dim3(256), 0, st, a);
    return;
  }
#endif
  constexpr int PRIO_DEFAULT = 0;
  const int prio = tune_env("MIMO_ATTN40_PRIO", PRIO_DEFAULT);
  if (prio == 1) hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 1>), grid, dim3(256), 0, st, a);
  else if (prio == 2) hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 2>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((attn40_kernel<DT, 4, 0, 0, 0>), grid, dim3(256), 0, st, a);
}

static int attention_impl(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                          const void* v, int64_t ldv, const void* k2, int64_t ldk2, const void* v2,
                          int64_t ldv2, void* out, int64_t ldo, int B, int Nq, int Nk, int Nk2,
                          int seg2_first_batch, int heads, int d, float scale, void* stream, int qk8) {
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
  if (qk8) {  // opt-in fp8 Q.K^T: the generic flash kernel with the K tile / Q fragments in e4m3
#define ATTN_LAUNCH8(DT, DD) hipLaunchKernelGGL((attn_kernel<DT, DD, false, true>), grid, dim3(256), 0, st, a)
    if (dtype == MIMO_F16) {
      switch (d) {
        case 40: ATTN_LAUNCH8(MIMO_F16, 40); break;
        case 80: ATTN_LAUNCH8(MIMO_F16, 80); break;
        case 160: ATTN_LAUNCH8(MIMO_F16, 160); break;
        default: return MIMO_EINVAL;
      }
    } else if (dtype == MIMO_BF16) {
      switch (d) {
        case 40: ATTN_LAUNCH8(MIMO_BF16, 40); break;
        case 80: ATTN_LAUNCH8(MIMO_BF16, 80); break;
        case 160: ATTN_LAUNCH8(MIMO_BF16, 160); break;
        default: return MIMO_EINVAL;
      }
    } else {
      return MIMO_EDTYPE;
    }
#undef ATTN_LAUNCH8
    MIMO_LAUNCH_CHECK();
    return MIMO_OK;
  }
#define ATTN_LAUNCH(DT, DD) hipLaunchKernelGGL((attn_kernel<DT, DD, false>), grid, dim3(256), 0, st, a)
  // d = 512: one head, no second segment, explicit scale (the VAE mid-block attention)
#define ATTN_LAUNCH512(DT)                                                                         \
  do {                                                                                             \
    if (heads != 1 || Nk2 > 0 || a.prescaled) return MIMO_EINVAL;                                  \
    hipLaunchKernelGGL((attn512_kernel<DT>), dim3((unsigned)((Nq + 31) / 32), (unsigned)B), dim3(256), 0, st, a); \
  } while (0)
#define ATTN_LAUNCH40(DT)                                                                          \
  do {                                                                                             \
    if (a.prescaled && !attn40_legacy())                                                           \
      attn40_launch<DT>(a, st);                                                                    \
    else if (a.prescaled) hipLaunchKernelGGL((attn_kernel<DT, 40, true>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((attn_kernel<DT, 40, false>), grid, dim3(256), 0, st, a);              \
  } while (0)
  if (dtype == MIMO_F16) {
    switch (d) {
      case 40: ATTN_LAUNCH40(MIMO_F16); break;
      case 64: ATTN_LAUNCH(MIMO_F16, 64); break;
      case 80: ATTN_LAUNCH(MIMO_F16, 80); break;
      case 160: ATTN_LAUNCH(MIMO_F16, 160); break;
      case 512: ATTN_LAUNCH512(MIMO_F16); break;
      default: return MIMO_EINVAL;
    }
  } else if (dtype == MIMO_BF16) {
    switch (d) {
      case 40: ATTN_LAUNCH40(MIMO_BF16); break;
      case 64: ATTN_LAUNCH(MIMO_BF16, 64); break;
      case 80: ATTN_LAUNCH(MIMO_BF16, 80); break;
      case 160: ATTN_LAUNCH(MIMO_BF16, 160); break;
      case 512: ATTN_LAUNCH512(MIMO_BF16); break;
      default: return MIMO_EINVAL;
    }
  } else {
    return MIMO_EDTYPE;
  }
#undef ATTN_LAUNCH
#undef ATTN_LAUNCH512
#undef ATTN_LAUNCH40
  MIMO_LAUNCH_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_attention(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                              const void* v, int64_t ldv, const void* k2, int64_t ldk2, const void* v2,
                              int64_t ldv2, void* out, int64_t ldo, int B, int Nq, int Nk, int Nk2,
                              int seg2_first_batch, int heads, int d, float scale, void* stream) {
  return attention_impl(dtype, q, ldq, k, ldk, v, ldv, k2, ldk2, v2, ldv2, out, ldo, B, Nq, Nk, Nk2, seg2_first_batch, heads, d,
                        scale, stream, 0);
}

extern "C" int mimo_attention_fp8qk(int dtype, const void* q, int64_t ldq, const void* k, int64_t ldk,
                                    const void* v, int64_t ldv, const void* k2, int64_t ldk2, const void* v2,
                                    int64_t ldv2, void* out, int64_t ldo, int B, int Nq, int Nk, int Nk2,
                                    int seg2_first_batch, int heads, int d, float scale, void* stream) {
  return attention_impl(dtype, q, ldq, k, ldk, v, ldv, k2, ldk2, v2, ldv2, out, ldo, B, Nq, Nk, Nk2, seg2_first_batch, heads, d,
                        scale, stream, 1);
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
  // the second form needs 16-byte-aligned rows everywhere (every call of the UNets qualifies); MIMO_TATTN_LEGACY=1 (tune
  // build only) keeps the first form for A/B timing
  const bool v2 = !(ldv & 7) && !(ldo & 7) && !(reinterpret_cast<uintptr_t>(v) & 15) && !(reinterpret_cast<uintptr_t>(out) & 15) &&
                  !(reinterpret_cast<uintptr_t>(q) & 15) && !(reinterpret_cast<uintptr_t>(k) & 15) && !tune_env("MIMO_TATTN_LEGACY", 0);
#define TATTN_LAUNCH(DT, DD)                                                                                          \
  do {                                                                                                                \
    if (v2) hipLaunchKernelGGL((temporal_attn2_kernel<DT, DD>), dim3((unsigned)nb), dim3(256), 0, st, a);             \
    else hipLaunchKernelGGL((temporal_attn_kernel<DT, DD>), dim3((unsigned)nb), dim3(256), 0, st, a);                 \
  } while (0)
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
