// Resident-grid forms of the three passes of the UNMASKED attention (the ViT's: no mask, not causal, no dropout;
// reference models/vit.py:72-84) -- round 6.  Device code shared with csrc/attn.hip lives in attn_common.h.
#include "attn_common.h"

namespace bq {

// ---- persistent grid (round 6) -------------------------------------------------------------------------------------------
// The ViT's attention at 512 x 512 (L = 1025 = 8 x 128 + the class token, B.H = 192) is 1536 full 128-row blocks plus 192
// one-row blocks: launched block by block, four to a CU, that is 1.5 rounds of full blocks (the second round runs at half
// occupancy), every block pays its own prologue (Q fragments, the first tile's round trip) and epilogue, and the one-row
// blocks keep a wave slot busy for 17 latency-bound tile turns.  Here the grid is resident -- 3 workgroups per CU -- and
// every workgroup walks a list of (head, query block) items as ONE stream of key tiles: the next item's first K / V tile
// and Q fragments are requested while the current item's last tile is consumed, so the LDS-DMA pipeline never drains;
// the items of a head run on workgroups of one XCD (blockIdx.x % 8: that L2 then holds the head's K / V once instead of
// eight L2s holding it); the single ragged row of a head -- when Lq = 128 n + 1 -- is done afterwards on plain vector
// arithmetic by the four waves of one workgroup (attn_fwd_tail_row: no tile pipeline to fill for one row).
struct PersistItem {
  int bh, qb;
};
__device__ __forceinline__ PersistItem persist_item(int slot, int nqb, int xcd) {
  const int hl = slot / nqb;
  return PersistItem{hl * 8 + xcd, slot - hl * nqb};
}

// one 16-B load of 8 bf16 as 4 packed pairs
__device__ __forceinline__ uint4 ld16(const __bf16 *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ float dot8_bf16(const uint4 a, const uint4 b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  float acc = 0.0f;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x), __builtin_bit_cast(bf16x2, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y), __builtin_bit_cast(bf16x2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z), __builtin_bit_cast(bf16x2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w), __builtin_bit_cast(bf16x2, b.w), acc, false);
  return acc;
}
// sum over the 8 lanes of a lane group (lane / 8): every lane of the group ends with the total
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}
// over the 8 lane groups of a wave (lanes of equal lane % 8)
__device__ __forceinline__ float across_groups_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8));
  v = fmaxf(v, __shfl_xor(v, 16));
  return xhalf_max(v);
}
__device__ __forceinline__ float across_groups_sum(float v) {
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  return xhalf_sum(v);
}
__device__ __forceinline__ void unpack8(const uint4 u, float *f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// ONE query row `q` of head (b, hd) against all Lk keys, no mask / dropout, on vector arithmetic: the four waves take the
// 64-key chunks round-robin; inside a wave, lane group gq = lane / 8 owns key 8 i + gq of the chunk and its 8 lanes own 8
// of the 64 dims each (one coalesced 128-B row per group, K and V loads of a chunk in flight together); online softmax per
// wave, the four partial states merged through `lds` (>= 4 * 66 floats; barrier-protected on both sides).
__device__ __forceinline__ void attn_fwd_tail_row(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                  const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                  float *__restrict__ LSE, const AttnDims &dm, int bh, int q, float *lds) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, gq = lane >> 3, sub = lane & 7;
  const int b = bh / dm.H, hd = bh % dm.H;
  const float c = dm.scale * 1.4426950408889634f;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs + 8 * sub;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs + 8 * sub;
  const uint4 qx = ld16(Q + b * dm.q_bs + hd * dm.q_hs + (long)q * dm.q_rs + 8 * sub);
  float m = -INFINITY, l = 0.0f, acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
  const int nch = (dm.Lk + 63) >> 6;
  for (int ch = wid; ch < nch; ch += AT_NW) {
    uint4 kx[8], vx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = min(ch * 64 + 8 * i + gq, dm.Lk - 1);
      kx[i] = ld16(Kb + (long)key * dm.k_rs);
      vx[i] = ld16(Vb + (long)key * dm.k_rs);
    }
    float sc[8], cm = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sc[i] = group8_sum(dot8_bf16(qx, kx[i])) * c;
      if (ch * 64 + 8 * i + gq >= dm.Lk) sc[i] = -INFINITY;
      cm = fmaxf(cm, sc[i]);
    }
    cm = across_groups_max(cm);   // (every chunk holds at least one valid key: finite)
    const float mnew = fmaxf(m, cm), alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
    l *= alpha;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] *= alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float p = __builtin_amdgcn_exp2f(sc[i] - mnew);
      // the matrix path rounds P to bf16 before P.V and sums the unrounded values: the same here
      const float pr = (float)(__bf16)p;
      l += p;
      float vf[8];
      unpack8(vx[i], vf);
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] = __builtin_fmaf(pr, vf[d], acc[d]);
    }
  }
  l = across_groups_sum(l);
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = across_groups_sum(acc[d]);
  __syncthreads();   // (the caller's tile images, which `lds` aliases, are no longer read)
  float *s_o = lds, *s_m = lds + AT_NW * 64, *s_l = s_m + AT_NW;
  if (gq == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) s_o[wid * 64 + 8 * sub + d] = acc[d];
    if (sub == 0) { s_m[wid] = m; s_l[wid] = l; }
  }
  __syncthreads();
  if (t < 64) {
    float mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < AT_NW; ++w) mstar = fmaxf(mstar, s_m[w]);
    float lt = 0.0f, o = 0.0f;
#pragma unroll
    for (int w = 0; w < AT_NW; ++w) {
      const float wg = s_m[w] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s_m[w] - mstar);
      lt += wg * s_l[w];
      o += wg * s_o[w * 64 + t];
    }
    O[b * dm.o_bs + hd * dm.o_hs + (long)q * dm.o_rs + t] = (__bf16)(o / lt);
    if (t == 0) LSE[(long)bh * dm.Lq + q] = mstar + __builtin_amdgcn_logf(lt);
  }
  __syncthreads();   // (`lds` may be rewritten by the caller's next row)
}

// nqb: query blocks per head that go through the tile pipeline (the full ones, + a ragged one of 2 .. 127 rows);
// tail: 1 when every head has ONE more row (Lq = 128 * nqb + 1) for attn_fwd_tail_row.  B.H must be a multiple of 8 and
// gridDim.x a multiple of 8 (workgroups w, w + 8, ... share an XCD and walk the heads w % 8, w % 8 + 8, ...).
// (a TEMPLATE on purpose, and every specialisation used by exactly ONE kernel: a non-template __device__ function that
// instantiates TileDma<> is analysed by the HOST pass, fails there on the device builtins and takes every later user of
// TileDma<> down with it -- "substitution failure"; so does the second kernel that calls one specialisation)
template <int UNUSED, bool PK>
__device__ __forceinline__ void attn_fwd_persist_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                      const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                      float *__restrict__ LSE, const AttnDims &dm, int nqb, int tail) {
  __shared__ __align__(16) unsigned char s_k[2][AT_KB * 128];
  __shared__ __align__(16) unsigned char s_v[2][AT_KB * 128];
  const float scale_log2e = dm.scale * 1.4426950408889634f;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, G8 = gridDim.x >> 3;
  const int nslots = (dm.B * dm.H >> 3) * nqb;
  const int nkt = (dm.Lk + AT_KB - 1) / AT_KB;
  int slot = blockIdx.x >> 3;
  if (slot < nslots) {
    PersistItem it = persist_item(slot, nqb, xcd);
    TileDma<true> tdma;
    bf16x8 qf[4];
    auto bases = [&](const PersistItem &i_) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      tdma.init(K + b * dm.k_bs + hd * dm.k_hs, dm.k_rs, V + b * dm.k_bs + hd * dm.k_hs, dm.k_rs, dm.Lk, lane, wid, AT_NW);
    };
    auto load_q = [&](const PersistItem &i_, bf16x8 (&dst)[4]) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      const int qr = min(i_.qb * AT_QB + wid * AT_QW + r, dm.Lq - 1);
      const __bf16 *Qrow = Q + b * dm.q_bs + hd * dm.q_hs + (long)qr * dm.q_rs;
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[s] = *reinterpret_cast<const bf16x8 *>(Qrow + 16 * s + 8 * h);
    };
    bases(it);
    load_q(it, qf);
    int g = 0;   // tiles consumed so far: tile g lives in image g & 1
    tdma.issue(0, s_k[0], s_v[0]);
    while (true) {
      const int snext = slot + G8;
      const bool more = snext < nslots;
      const PersistItem nx = persist_item(more ? snext : slot, nqb, xcd);
      bf16x8 qn[4];
      f32x16 o0 = {0}, o1 = {0};
      float m = -INFINITY, lsum = 0.0f;
      const int q0 = it.qb * AT_QB + wid * AT_QW;
      const bool active = q0 < dm.Lq;   // wave-uniform: a wave of a ragged block with nothing to do still stages
      for (int kt = 0; kt < nkt - 1; ++kt, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tdma.issue(kt + 1, s_k[(g + 1) & 1], s_v[(g + 1) & 1]);
        if (active)
          fwd_tile<true, false, false, PK>(s_k[g & 1], s_v[g & 1], qf, dm, nullptr, scale_log2e, 0u, it.bh, q0 + r, kt, false, r, h, o0, o1, m, lsum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {   // the next item's first tile and Q fragments travel under this item's last tile
        bases(nx);
        tdma.issue(0, s_k[(g + 1) & 1], s_v[(g + 1) & 1]);
        load_q(nx, qn);
      }
      if (active)
        fwd_tile<true, true, false, PK>(s_k[g & 1], s_v[g & 1], qf, dm, nullptr, scale_log2e, 0u, it.bh, q0 + r, nkt - 1, true, r, h, o0, o1, m, lsum);
      ++g;
      {
        const float l = xhalf_sum(lsum);
        const float inv = 1.0f / l;
        const int q = q0 + r;
        if (q < dm.Lq) {
          const int b = it.bh / dm.H, hd = it.bh % dm.H;
          __bf16 *Orow = O + b * dm.o_bs + hd * dm.o_hs + (long)q * dm.o_rs;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bf16x4 w0, w1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              w0[j] = (__bf16)(o0[4 * g4 + j] * inv);
              w1[j] = (__bf16)(o1[4 * g4 + j] * inv);
            }
            *reinterpret_cast<bf16x4 *>(Orow + 8 * g4 + 4 * h) = w0;
            *reinterpret_cast<bf16x4 *>(Orow + 32 + 8 * g4 + 4 * h) = w1;
          }
          if (h == 0) LSE[(long)it.bh * dm.Lq + q] = m + __builtin_amdgcn_logf(l);
        }
      }
      if (!more) break;
      slot = snext;
      it = nx;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = qn[s];
    }
  }
  if (tail) {
    const int BH = dm.B * dm.H;
    for (int bh = blockIdx.x; bh < BH; bh += gridDim.x)
      attn_fwd_tail_row(Q, K, V, O, LSE, dm, bh, dm.Lq - 1, reinterpret_cast<float *>(&s_k[0][0]));
  }
}

// (a __device__ body: see attn_fwd_kernel)
template <int MINW, bool PK>
__global__ __launch_bounds__(256, MINW) void attn_fwd_persist_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                               const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                               float *__restrict__ LSE, AttnDims dm, int nqb, int tail) {
  attn_fwd_persist_body<MINW, PK>(Q, K, V, O, LSE, dm, nqb, tail);
}

// ---- the dQ pass on a resident grid (see attn_fwd_persist_body) -----------------------------------------------------------
// dQ of ONE query row on vector arithmetic (the 128 n + 1-th row of every head): the four waves take the 64-key chunks
// round-robin, lane group gq owns key 8 i + gq, its 8 lanes 8 dims each; s = q.k and dp = dO.v per key (two 8-lane sums),
// ds = p (dp - delta), dq += ds k; partial dq merged through `lds` (>= 4 * 64 floats).  Writes DELTA[q] too.
__device__ __forceinline__ void attn_bwd_dq_tail_row(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                     const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                     const float *__restrict__ LSE, const __bf16 *__restrict__ O,
                                                     float *__restrict__ DELTA, __bf16 *__restrict__ dQ, const BwdDims &dm,
                                                     int bh, int q, float *lds) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, gq = lane >> 3, sub = lane & 7;
  const int b = bh / dm.H, hd = bh % dm.H;
  const float c = dm.scale * 1.4426950408889634f;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs + 8 * sub;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs + 8 * sub;
  const uint4 qx = ld16(Q + b * dm.q_bs + hd * dm.q_hs + (long)q * dm.q_rs + 8 * sub);
  const uint4 gx = ld16(dO + b * dm.o_bs + hd * dm.o_hs + (long)q * dm.o_rs + 8 * sub);
  const uint4 ox = ld16(O + (((long)b * dm.Lq + q) * dm.H + hd) * AT_D + 8 * sub);
  const float lse = LSE[(long)bh * dm.Lq + q];
  float gf_[8], of_[8], delta = 0.0f;
  unpack8(gx, gf_);
  unpack8(ox, of_);
#pragma unroll
  for (int d = 0; d < 8; ++d) delta += of_[d] * gf_[d];
  delta = group8_sum(delta);
  if (t == 0) DELTA[(long)bh * dm.Lq + q] = delta;
  float acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = 0.0f;
  const int nch = (dm.Lk + 31) >> 5;   // (32-key chunks: 4 + 4 loads in flight per lane -- 8 + 8 spill beside the tile loop's registers)
  for (int ch = wid; ch < nch; ch += AT_NW) {
    uint4 kx[4], vx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = min(ch * 32 + 8 * i + gq, dm.Lk - 1);
      kx[i] = ld16(Kb + (long)key * dm.k_rs);
      vx[i] = ld16(Vb + (long)key * dm.k_rs);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sdot = group8_sum(dot8_bf16(qx, kx[i])), dp = group8_sum(dot8_bf16(gx, vx[i]));
      float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sdot, c, -lse));
      if (ch * 32 + 8 * i + gq >= dm.Lk) p = 0.0f;
      const float ds = (float)(__bf16)(p * (dp - delta));   // (the matrix path rounds dS to bf16 before dS.K)
      float kf_[8];
      unpack8(kx[i], kf_);
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] = __builtin_fmaf(ds, kf_[d], acc[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = across_groups_sum(acc[d]);
  __syncthreads();
  if (gq == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) lds[wid * 64 + 8 * sub + d] = acc[d];
  }
  __syncthreads();
  if (t < 64)
    dQ[b * dm.q_bs + hd * dm.q_hs + (long)q * dm.q_rs + t] =
        (__bf16)(((lds[t] + lds[64 + t]) + (lds[128 + t] + lds[192 + t])) * dm.scale);
  __syncthreads();
}

template <int UNUSED>
__device__ __forceinline__ void attn_bwd_dq_persist_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                         const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                         const float *__restrict__ LSE, const __bf16 *__restrict__ O,
                                                         float *__restrict__ DELTA, __bf16 *__restrict__ dQ,
                                                         const BwdDims &dm, int nqb, int tail) {
  __shared__ __align__(16) unsigned char s_k[2][AT_KB * 128];
  __shared__ __align__(16) unsigned char s_v[2][AT_KB * 128];
  const float scale = dm.scale, c = scale * 1.4426950408889634f;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, G8 = gridDim.x >> 3;
  const int nslots = (dm.B * dm.H >> 3) * nqb;
  const int nkt = (dm.Lk + AT_KB - 1) / AT_KB;
  int slot = blockIdx.x >> 3;
  if (slot < nslots) {
    PersistItem it = persist_item(slot, nqb, xcd);
    TileDma<true> tdma;
    bf16x8 qf[4], gf[4];
    auto bases = [&](const PersistItem &i_) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      tdma.init(K + b * dm.k_bs + hd * dm.k_hs, dm.k_rs, V + b * dm.k_bs + hd * dm.k_hs, dm.k_rs, dm.Lk, lane, wid, AT_NW);
    };
    auto load_qg = [&](const PersistItem &i_, bf16x8 (&dq_)[4], bf16x8 (&dg_)[4]) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      const int qr = min(i_.qb * AT_QB + wid * AT_QW + r, dm.Lq - 1);
      const __bf16 *Qrow = Q + b * dm.q_bs + hd * dm.q_hs + (long)qr * dm.q_rs;
      const __bf16 *Grow = dO + b * dm.o_bs + hd * dm.o_hs + (long)qr * dm.o_rs;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        dq_[s] = *reinterpret_cast<const bf16x8 *>(Qrow + 16 * s + 8 * h);
        dg_[s] = *reinterpret_cast<const bf16x8 *>(Grow + 16 * s + 8 * h);
      }
    };
    // lse and delta[q] = rowsum(dO o O) of this lane's query row (O: the forward's contiguous (B, Lq, H, 64) output)
    auto row_scalars = [&](const PersistItem &i_, const bf16x8 (&g_)[4], float &lse, float &delta) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      const int q0 = i_.qb * AT_QB + wid * AT_QW, qr = min(q0 + r, dm.Lq - 1);
      lse = LSE[(long)i_.bh * dm.Lq + qr];
      const __bf16 *Orow = O + (((long)b * dm.Lq + qr) * dm.H + hd) * AT_D;
      float d_ = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(Orow + 16 * s + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) d_ += (float)ov[j] * (float)g_[s][j];
      }
      delta = xhalf_sum(d_);
      if (h == 0 && q0 + r < dm.Lq) DELTA[(long)i_.bh * dm.Lq + q0 + r] = delta;
    };
    bases(it);
    load_qg(it, qf, gf);
    int g = 0;
    tdma.issue(0, s_k[0], s_v[0]);
    float lse, delta;
    row_scalars(it, gf, lse, delta);
    while (true) {
      const int snext = slot + G8;
      const bool more = snext < nslots;
      const PersistItem nx = persist_item(more ? snext : slot, nqb, xcd);
      f32x16 a0 = {0}, a1 = {0};
      const int q0 = it.qb * AT_QB + wid * AT_QW;
      const bool active = q0 < dm.Lq;
      for (int kt = 0; kt < nkt - 1; ++kt, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tdma.issue(kt + 1, s_k[(g + 1) & 1], s_v[(g + 1) & 1]);
        if (active)
          dq_tile<true, false>(s_k[g & 1], s_v[g & 1], qf, gf, dm, nullptr, c, scale, lse, delta, 0u, it.bh, q0 + r, kt, false, r, h, a0, a1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {   // (the next item's first tile travels under this item's last; its Q / dO fragments are fetched after the
        bases(nx);  //  epilogue -- held beside the live ones they would spill 25 registers)
        tdma.issue(0, s_k[(g + 1) & 1], s_v[(g + 1) & 1]);
      }
      if (active)
        dq_tile<true, true>(s_k[g & 1], s_v[g & 1], qf, gf, dm, nullptr, c, scale, lse, delta, 0u, it.bh, q0 + r, nkt - 1, true, r, h, a0, a1);
      ++g;
      if (q0 + r < dm.Lq) {
        const int b = it.bh / dm.H, hd = it.bh % dm.H;
        store_T(dQ + b * dm.q_bs + hd * dm.q_hs + (long)(q0 + r) * dm.q_rs, a0, a1, h, scale);
      }
      if (!more) break;
      slot = snext;
      it = nx;
      load_qg(it, qf, gf);
      row_scalars(it, gf, lse, delta);
    }
  }
  if (tail) {
    const int BH = dm.B * dm.H;
    for (int bh = blockIdx.x; bh < BH; bh += gridDim.x)
      attn_bwd_dq_tail_row(Q, K, V, dO, LSE, O, DELTA, dQ, dm, bh, dm.Lq - 1, reinterpret_cast<float *>(&s_k[0][0]));
  }
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq_persist_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                                  const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                                  const float *__restrict__ LSE, const __bf16 *__restrict__ O,
                                                                  float *__restrict__ DELTA, __bf16 *__restrict__ dQ, BwdDims dm,
                                                                  int nqb, int tail) {
  attn_bwd_dq_persist_body<MINW>(Q, K, V, dO, LSE, O, DELTA, dQ, dm, nqb, tail);
}

// ---- the dK/dV pass on a resident grid (see attn_fwd_persist_body) -------------------------------------------------------
// dK / dV of ONE key (the 128 n + 1-th key of every head) on vector arithmetic: the four waves take 32-query chunks
// round-robin, lane group gq owns query 8 i + gq, its 8 lanes 8 dims each.  DELTA must be complete (the dQ pass ran before).
__device__ __forceinline__ void attn_bwd_dkv_tail_key(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                      const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                      const float *__restrict__ LSE, const float *__restrict__ DELTA,
                                                      __bf16 *__restrict__ dK, __bf16 *__restrict__ dV, const BwdDims &dm,
                                                      int bh, int key, float *lds) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, gq = lane >> 3, sub = lane & 7;
  const int b = bh / dm.H, hd = bh % dm.H;
  const float c = dm.scale * 1.4426950408889634f;
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs + 8 * sub;
  const __bf16 *Gb = dO + b * dm.o_bs + hd * dm.o_hs + 8 * sub;
  const long koff = b * dm.k_bs + hd * dm.k_hs + (long)key * dm.k_rs;
  const uint4 kx = ld16(K + koff + 8 * sub), vx = ld16(V + koff + 8 * sub);
  const float *lseb = LSE + (long)bh * dm.Lq, *delb = DELTA + (long)bh * dm.Lq;
  float ak[8], av[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) ak[d] = av[d] = 0.0f;
  const int nch = (dm.Lq + 31) >> 5;
  for (int ch = wid; ch < nch; ch += AT_NW) {
    uint4 qx[4], gx[4];
    float lv[4], dl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qq = min(ch * 32 + 8 * i + gq, dm.Lq - 1);
      qx[i] = ld16(Qb + (long)qq * dm.q_rs);
      gx[i] = ld16(Gb + (long)qq * dm.o_rs);
      lv[i] = lseb[qq];
      dl[i] = delb[qq];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sdot = group8_sum(dot8_bf16(qx[i], kx)), dp = group8_sum(dot8_bf16(gx[i], vx));
      float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sdot, c, -lv[i]));
      if (ch * 32 + 8 * i + gq >= dm.Lq) p = 0.0f;
      const float pr = (float)(__bf16)p, ds = (float)(__bf16)(p * (dp - dl[i]));   // (bf16 operands, as on the matrix path)
      float qf_[8], gf_[8];
      unpack8(qx[i], qf_);
      unpack8(gx[i], gf_);
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        av[d] = __builtin_fmaf(pr, gf_[d], av[d]);
        ak[d] = __builtin_fmaf(ds, qf_[d], ak[d]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    ak[d] = across_groups_sum(ak[d]);
    av[d] = across_groups_sum(av[d]);
  }
  __syncthreads();
  if (gq == 0) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      lds[wid * 64 + 8 * sub + d] = ak[d];
      lds[256 + wid * 64 + 8 * sub + d] = av[d];
    }
  }
  __syncthreads();
  if (t < 64) {
    dK[koff + t] = (__bf16)(((lds[t] + lds[64 + t]) + (lds[128 + t] + lds[192 + t])) * dm.scale);
    dV[koff + t] = (__bf16)((lds[256 + t] + lds[320 + t]) + (lds[384 + t] + lds[448 + t]));
  }
  __syncthreads();
}

// nkb: key blocks per head on the tile pipeline; tail: one more key per head (Lk = 128 nkb + 1) for attn_bwd_dkv_tail_key.
// Needs at least two query tiles (Lq > 64).
template <int UNUSED>
__device__ __forceinline__ void attn_bwd_dkv_persist_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                          const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                          const float *__restrict__ LSE, const float *__restrict__ DELTA,
                                                          __bf16 *__restrict__ dK, __bf16 *__restrict__ dV,
                                                          const BwdDims &dm, int nkb, int tail) {
  __shared__ __align__(16) unsigned char s_q[2][AT_KB * 128];
  __shared__ __align__(16) unsigned char s_g[2][AT_KB * 128];
  __shared__ __align__(16) float s_lse[2][AT_KB];
  __shared__ __align__(16) float s_del[2][AT_KB];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, G8 = gridDim.x >> 3;
  const int nslots = (dm.B * dm.H >> 3) * nkb;
  const int nqt = (dm.Lq + AT_KB - 1) / AT_KB;
  const float scale = dm.scale, c = scale * 1.4426950408889634f;
  int slot = blockIdx.x >> 3;
  if (slot < nslots) {
    PersistItem it = persist_item(slot, nkb, xcd);
    TileDma<false> tdma;
    auto bases = [&](const PersistItem &i_) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      tdma.init(Q + b * dm.q_bs + hd * dm.q_hs, dm.q_rs, dO + b * dm.o_bs + hd * dm.o_hs, dm.o_rs, dm.Lq, lane, wid, AT_NW);
    };
    float rl = 0.f, rd = 0.f;
    auto fetch = [&](int bh, int qt) {   // the 64 row scalars of a tile into two registers of the first wave
      if (t < AT_KB) {
        const long qq = (long)bh * dm.Lq + min(qt * AT_KB + t, dm.Lq - 1);
        rl = LSE[qq];
        rd = DELTA[qq];
      }
    };
    auto commit = [&](int buf) {
      if (t < AT_KB) { s_lse[buf][t] = rl; s_del[buf][t] = rd; }
    };
    bf16x8 kf[4], vf[4];
    auto load_kv = [&](const PersistItem &i_) {
      const int b = i_.bh / dm.H, hd = i_.bh % dm.H;
      const int kr = min(i_.qb * AT_QB + wid * AT_QW + r, dm.Lk - 1);
      const long off = b * dm.k_bs + hd * dm.k_hs + (long)kr * dm.k_rs;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8 *>(K + off + 16 * s + 8 * h);
        vf[s] = *reinterpret_cast<const bf16x8 *>(V + off + 16 * s + 8 * h);
      }
    };
    bases(it);
    fetch(it.bh, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    commit(0);
    tdma.issue(0, s_q[0], s_g[0]);
    fetch(it.bh, 1);
    load_kv(it);
    int g = 0;   // tiles consumed so far: tile g lives in image g & 1
    while (true) {
      const int snext = slot + G8;
      const bool more = snext < nslots;
      const PersistItem nx = persist_item(more ? snext : slot, nkb, xcd);
      f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
      const int k0 = it.qb * AT_QB + wid * AT_QW;
      const bool active = k0 < dm.Lk;
      for (int qt = 0; qt < nqt - 1; ++qt, ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        commit((g + 1) & 1);   // (tile qt + 1's scalars: fetched one turn ago)
        tdma.issue(qt + 1, s_q[(g + 1) & 1], s_g[(g + 1) & 1]);
        if (qt + 2 < nqt) fetch(it.bh, qt + 2);
        else if (more) fetch(nx.bh, 0);
        if (active)
          dkv_tile<true, false>(s_q[g & 1], s_g[g & 1], s_lse[g & 1], s_del[g & 1], kf, vf, dm, c, 0.0f, 0u, it.bh, k0 + r, qt,
                                false, r, h, dk0, dk1, dv0, dv1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (more) {   // the next item's first tile travels under this item's last one
        commit((g + 1) & 1);
        bases(nx);
        tdma.issue(0, s_q[(g + 1) & 1], s_g[(g + 1) & 1]);
        fetch(nx.bh, 1);
      }
      if (active)
        dkv_tile<true, true>(s_q[g & 1], s_g[g & 1], s_lse[g & 1], s_del[g & 1], kf, vf, dm, c, 0.0f, 0u, it.bh, k0 + r, nqt - 1,
                             true, r, h, dk0, dk1, dv0, dv1);
      ++g;
      if (k0 + r < dm.Lk) {
        const int b = it.bh / dm.H, hd = it.bh % dm.H;
        const long off = b * dm.k_bs + hd * dm.k_hs + (long)(k0 + r) * dm.k_rs;
        store_T(dK + off, dk0, dk1, h, scale);
        store_T(dV + off, dv0, dv1, h, 1.0f);
      }
      if (!more) break;
      slot = snext;
      it = nx;
      load_kv(it);
    }
  }
  if (tail) {
    const int BH = dm.B * dm.H;
    for (int bh = blockIdx.x; bh < BH; bh += gridDim.x)
      attn_bwd_dkv_tail_key(Q, K, V, dO, LSE, DELTA, dK, dV, dm, bh, dm.Lk - 1, reinterpret_cast<float *>(&s_q[0][0]));
  }
}

template <int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dkv_persist_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                                   const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                                   const float *__restrict__ LSE, const float *__restrict__ DELTA,
                                                                   __bf16 *__restrict__ dK, __bf16 *__restrict__ dV, BwdDims dm,
                                                                   int nkb, int tail) {
  attn_bwd_dkv_persist_body<MINW>(Q, K, V, dO, LSE, DELTA, dK, dV, dm, nkb, tail);
}

int attn_fwd_persist_launch(const void *Q, const void *K, const void *V, void *O, float *LSE, const AttnDims &dm, int nqb,
                            int tail, int slots, hipStream_t st) {
  // (12 KB of unused dynamic LDS on top of the 32 KB of tile images: a FOURTH workgroup must not fit a CU -- the dispatcher
  // would stack four on some CUs and leave two on others)
  if (tail & 2)   // (measurement: the scalar softmax arithmetic)
    hipLaunchKernelGGL((attn_fwd_persist_kernel<3, false>), dim3(slots), dim3(256), 12288, st, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (__bf16 *)O, LSE, dm, nqb, tail & 1);
  else
    hipLaunchKernelGGL((attn_fwd_persist_kernel<3, true>), dim3(slots), dim3(256), 12288, st, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (__bf16 *)O, LSE, dm, nqb, tail & 1);
  return check_launch("attn_fwd_persist");
}

int attn_bwd_dq_persist_launch(const void *Q, const void *K, const void *V, const void *dO, const float *LSE, const void *O,
                               float *DELTA, void *dQ, const BwdDims &dm, int nqb, int tail, int slots, hipStream_t st) {
  hipLaunchKernelGGL((attn_bwd_dq_persist_kernel<3>), dim3(slots), dim3(256), 12288, st, (const __bf16 *)Q, (const __bf16 *)K,
                     (const __bf16 *)V, (const __bf16 *)dO, LSE, (const __bf16 *)O, DELTA, (__bf16 *)dQ, dm, nqb, tail);
  return check_launch("attn_bwd_dq_persist");
}

int attn_bwd_dkv_persist_launch(const void *Q, const void *K, const void *V, const void *dO, const float *LSE,
                                const float *DELTA, void *dK, void *dV, const BwdDims &dm, int nkb, int tail, int slots,
                                hipStream_t st) {
  // (2 workgroups per CU: 24 KB of unused dynamic LDS keeps a third out)
  hipLaunchKernelGGL((attn_bwd_dkv_persist_kernel<2>), dim3(slots), dim3(256), 24576, st, (const __bf16 *)Q, (const __bf16 *)K,
                     (const __bf16 *)V, (const __bf16 *)dO, LSE, DELTA, (__bf16 *)dK, (__bf16 *)dV, dm, nkb, tail);
  return check_launch("attn_bwd_dkv_persist");
}

}  // namespace bq
