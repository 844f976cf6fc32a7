// Fused (non-materialising) multi-head attention for gfx950, head_dim 64, bf16 in / fp32 accumulate.
// Replaces the  q@k^T -> *scale -> softmax -> @v  chain of the reference's ViT attention
// (models/vit.py:75-83), which writes a (B,12,P,P) fp32 probability tensor per layer (807 MB at
// 512x512).  Forward here; the backward kernels live below it.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16; operand maps from cdna_hip_programming.md §3):
//   S^T[key][q] = K . Q^T       A = K rows from LDS (ds_read_b128), B = Q fragments held in registers.
//                               Result: lane = one query column, its 16 registers = 16 keys of the block,
//                               so the softmax max/sum are in-lane loops + ONE permlane32_swap.
//   O^T[d][q]  += V^T . P^T     B = the probabilities straight from the S^T accumulator registers (converted
//                               to bf16, no lane movement); A = V^T rows from an LDS image of the
//                               pre-transposed V ([d][key], key-contiguous) read in the k-order the
//                               accumulator registers have (16s + 8(j>>2) + 4h + (j&3)).
//                               Result: lane = query column again, so the online-softmax rescale of O is a
//                               per-lane scalar multiply.
// Workgroup = 4 waves = 128 query rows of one (batch, head); K / V^T tiles of 64 keys are staged through
// LDS once per workgroup (XOR-swizzled 16-B chunks), the next tile's global loads are in flight while the
// current one is consumed.
#include "attn_common.h"

namespace bq {

// MODE 0: every option at run time; 1: PLAIN (no mask / causal / dropout), last tile peeled; 2: PLAIN + EARLY scores
// (A ragged tail of 1 .. 32 queries -- L = 1025 = 8 x 128 + the CLS token -- costs a whole extra workgroup per (batch,
// head) that keeps one wave busy for the full key loop: 14 % of the launch at B = 64, tools/bench_attn_shapes.py L = 1024
// vs 1025.  Folding the tail into a FIFTH wave of query block 0 was built and measured in round 3: 320-thread workgroups
// drop the CU from four resident workgroups to three and every shape ran 10-50 % slower; not kept.)
template <int MODE>
__device__ __forceinline__ void attn_fwd_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                              const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                              float *__restrict__ LSE, const AttnDims &dm) {
  // two LDS images per operand: tile kt+1 lands in the other image while tile kt is consumed => ONE barrier per tile
  __shared__ __align__(16) unsigned char s_k[2][AT_KB * 128];
  __shared__ __align__(16) unsigned char s_v[2][AT_KB * 128];
  const float scale_log2e = dm.scale * 1.4426950408889634f;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / dm.H, hd = bh % dm.H;
  const unsigned seed = eff_seed(dm);
  const int q0 = blockIdx.x * AT_QB + wid * AT_QW;  // first query row of this wave
  const int live = attn_live_waves(dm.Lq, blockIdx.x * AT_QB);
  if (wid >= live) return;
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs;
  const float *mrow = dm.mask ? dm.mask + (long)b * dm.Lkp : nullptr;

  // Q fragments (B operand of S^T = K.Q^T): lane (q = r, h) holds Q[q][16*step + 8h + j]
  bf16x8 qf[4];
  {
    const int qr = min(q0 + r, dm.Lq - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(Qb + (long)qr * dm.q_rs + 16 * s + 8 * h);
  }
  f32x16 o0 = {0}, o1 = {0};
  float m = -INFINITY, lsum = 0.0f;
  const int nkt = (dm.Lk + AT_KB - 1) / AT_KB;
  TileDma<true> tdma;   // K / V tiles by LDS-DMA
  tdma.init(Kb, dm.k_rs, Vb, dm.k_rs, dm.Lk, lane, wid, live);
  auto dma = [&](int kt) { tdma.issue(kt, s_k[kt & 1], s_v[kt & 1]); };
  // before tile kt: its DMAs (issued one tile earlier) have landed for every wave and every wave has finished reading the
  // other image (tile kt - 1), which tile kt + 1 then overwrites while tile kt is consumed
  auto turn = [&](int kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) dma(kt + 1);
  };
  dma(0);
  const bool active = q0 < dm.Lq;  // wave-uniform: a wave whose 32 queries are all past the end only helps staging
  if (MODE == 0) {
    for (int kt = 0; kt < nkt; ++kt) {
      turn(kt);
      if (active)
        fwd_tile(s_k[kt & 1], s_v[kt & 1], qf, dm, mrow, scale_log2e, seed, bh, q0 + r, kt, kt == nkt - 1, r, h, o0, o1, m, lsum);
    }
  } else {
    for (int kt = 0; kt < nkt - 1; ++kt) {
      turn(kt);
      if (active)
        fwd_tile<true, false, MODE == 2>(s_k[kt & 1], s_v[kt & 1], qf, dm, nullptr, scale_log2e, seed, bh, q0 + r, kt, false, r, h, o0, o1, m, lsum);
    }
    turn(nkt - 1);
    if (active)
      fwd_tile<true, true, MODE == 2>(s_k[(nkt - 1) & 1], s_v[(nkt - 1) & 1], qf, dm, nullptr, scale_log2e, seed, bh, q0 + r, nkt - 1, true, r, h, o0, o1, m, lsum);
  }
  // epilogue: O[q][d] = O^T[d][q] / l ; LSE[q] = m + log2(l)   (log2 domain, scale folded in)
  const float l = xhalf_sum(lsum);
  const float inv = 1.0f / l;
  const int q = q0 + r;
  if (q < dm.Lq) {
    __bf16 *Orow = O + b * dm.o_bs + hd * dm.o_hs + (long)q * dm.o_rs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 w0, w1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w0[j] = (__bf16)(o0[4 * g + j] * inv);
        w1[j] = (__bf16)(o1[4 * g + j] * inv);
      }
      *reinterpret_cast<bf16x4 *>(Orow + 8 * g + 4 * h) = w0;
      *reinterpret_cast<bf16x4 *>(Orow + 32 + 8 * g + 4 * h) = w1;
    }
    if (h == 0) LSE[(long)bh * dm.Lq + q] = m + __builtin_amdgcn_logf(l);  // v_log_f32 = log2
  }
}

// (the body is a __device__ function: a __global__ body that instantiates TileDma<> loses its host launch stub -- the host
// pass fails on the device builtins inside the template without a diagnostic)
template <int MINW, int MODE = 0>
__global__ __launch_bounds__(256, MINW) void attn_fwd_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                       const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                       float *__restrict__ LSE, AttnDims dm) {
  attn_fwd_body<MODE>(Q, K, V, O, LSE, dm);
}

// (Round 2's second-generation PLAIN forward -- Q pre-multiplied by scale * log2(e) in bf16, the softmax reference subtracted
// on the matrix pipe by a fifth k-step, a lagging reference -- ran 10 % faster and doubled the output error / quadrupled
// the gradient error through the LSE; it was opt-in, never the default, and was removed in round 3.)
// Lq <= 32 (the text queries of the twin cross-attention: 20 tokens against 1045 image / 276 object keys): one
// workgroup per (batch, head) would keep ONE wave busy for ceil(Lk / 64) serial tiles.  Here the four waves share the
// 32 queries and take every fourth key tile each (own LDS images, own running max / sum / O^T), and the four partial
// softmax states are merged through LDS at the end:  O = sum_w 2^(m_w - m*) O_w / sum_w 2^(m_w - m*) l_w.
__device__ __forceinline__ void attn_fwd_narrow_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                     const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                     float *__restrict__ LSE, const AttnDims &dm, const int bh) {
  // every wave owns one K and one V image; the merge buffer s_o (33 KB) ALIASES them (it is written after the key loop,
  // behind a workgroup barrier): 65 KB instead of 100 KB of LDS => TWO workgroups per CU -- the pair launch's 2 x 192
  // workgroups are resident at once (they were 1.5 rounds of one per CU) and a SIMD has a second wave to issue from
  __shared__ __align__(16) unsigned char s_kv[2 * AT_NW * AT_KB * 128];
  __shared__ float s_m[AT_NW][32], s_l[AT_NW][32];
  unsigned char (*s_k)[AT_KB * 128] = reinterpret_cast<unsigned char (*)[AT_KB * 128]>(s_kv);
  unsigned char (*s_v)[AT_KB * 128] = reinterpret_cast<unsigned char (*)[AT_KB * 128]>(s_kv + AT_NW * AT_KB * 128);
  float (*s_o)[AT_D][33] = reinterpret_cast<float (*)[AT_D][33]>(s_kv);
  static_assert(sizeof(float) * AT_NW * AT_D * 33 <= 2 * AT_NW * AT_KB * 128, "merge buffer fits in the tile images");
  const float scale_log2e = dm.scale * 1.4426950408889634f;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int b = bh / dm.H, hd = bh % dm.H;
  const unsigned seed = eff_seed(dm);
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs;
  const float *mrow = dm.mask ? dm.mask + (long)b * dm.Lkp : nullptr;
  bf16x8 qf[4];
  {
    const int qr = min(r, dm.Lq - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(Qb + (long)qr * dm.q_rs + 16 * s + 8 * h);
  }
  f32x16 o0 = {0}, o1 = {0};
  float m = -INFINITY, lsum = 0.0f;
  const __bf16 *K2b = dm.K2 ? dm.K2 + b * dm.k2_bs + hd * dm.k2_hs : nullptr;  // second key/value segment (AttnDims)
  const __bf16 *V2b = dm.V2 ? dm.V2 + b * dm.k2_bs + hd * dm.k2_hs : nullptr;
  const int nkt = dm.nkt1 + (dm.Lk2 + AT_KB - 1) / AT_KB;
  // a wave stages, consumes and re-stages ITS OWN images: no workgroup barrier inside the loop (a wave's LDS operations
  // complete in order; the wave-level wait + barrier makes the lanes' stores visible to each other), and the NEXT tile's
  // global loads are in flight while this one is consumed
  {
    uint4 kr[8], vr[8];
    int kt = wid;
    KeyTile nx_ = key_tile(dm, Kb, Vb, K2b, V2b, min(kt, nkt - 1));
    if (kt < nkt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kr[j] = stage_load(nx_.k, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
        vr[j] = stage_load(nx_.v, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
      }
    }
    for (; kt < nkt; kt += AT_NW) {
      const KeyTile kt_ = nx_;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        stage_store(s_k[wid], lane + 64 * j, kr[j]);
        stage_store(s_v[wid], lane + 64 * j, vr[j]);
      }
      if (kt + AT_NW < nkt) {
        nx_ = key_tile(dm, Kb, Vb, K2b, V2b, kt + AT_NW);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kr[j] = stage_load(nx_.k, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
          vr[j] = stage_load(nx_.v, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      fwd_tile(s_k[wid], s_v[wid], qf, dm, mrow, scale_log2e, seed, bh, r, kt, kt_.last, r, h, o0, o1, m, lsum, kt_.kend);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (this tile's fragment reads are done before it is overwritten)
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();   // every wave has finished with its images: the merge buffer aliases them
  const float lw = xhalf_sum(lsum);
  if (h == 0) { s_m[wid][r] = m; s_l[wid][r] = lw; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s_o[wid][crow(i, h)][r] = o0[i];
    s_o[wid][32 + crow(i, h)][r] = o1[i];
  }
  __syncthreads();
  const int q = t & 31, d0 = (t >> 5) * 8;
  float mstar = -INFINITY;
#pragma unroll
  for (int w = 0; w < AT_NW; ++w) mstar = fmaxf(mstar, s_m[w][q]);
  float wgt[AT_NW], l = 0.0f;
#pragma unroll
  for (int w = 0; w < AT_NW; ++w) {
    wgt[w] = s_m[w][q] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(s_m[w][q] - mstar);
    l += wgt[w] * s_l[w][q];
  }
  if (q < dm.Lq) {
    const float inv = 1.0f / l;
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float acc = 0.0f;
#pragma unroll
      for (int w = 0; w < AT_NW; ++w) acc += wgt[w] * s_o[w][d0 + j][q];
      out[j] = (__bf16)(acc * inv);
    }
    *reinterpret_cast<bf16x8 *>(O + b * dm.o_bs + hd * dm.o_hs + (long)q * dm.o_rs + d0) = out;
    if (d0 == 0) LSE[(long)bh * dm.Lq + q] = mstar + __builtin_amdgcn_logf(l);
  }
}


__global__ __launch_bounds__(256) void attn_fwd_narrow_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                              const __bf16 *__restrict__ V, __bf16 *__restrict__ O,
                                                              float *__restrict__ LSE, AttnDims dm) {
  attn_fwd_narrow_body(Q, K, V, O, LSE, dm, blockIdx.y);
}

// Two independent narrow attentions in ONE launch (blockIdx.z = which): the two cross-attentions of a twin level (text
// queries over cat(image tokens, 3D states) and over cat(object tokens, 2D states): 1045 and 276 keys) are latency-bound
// launches of B.H workgroups each; side by side the short one runs under the long one.
struct AttnPair {
  const __bf16 *Q[2], *K[2], *V[2], *dO[2], *O[2];
  const float *LSE[2];
  float *LSEw[2], *DELTA[2];
  __bf16 *out[2], *dK[2], *dV[2];   // out: O (forward) or dQ (backward)
  AttnDims dm[2];
};

__global__ __launch_bounds__(256, 2) void attn_fwd_narrow_pair_kernel(const AttnPair a) {
  const int g = blockIdx.z;
  attn_fwd_narrow_body(a.Q[g], a.K[g], a.V[g], a.out[g], a.LSEw[g], a.dm[g], blockIdx.y);
}

// =====================================================================================================
// Backward.  With P = softmax(S), S = scale * Q K^T, delta[q] = rowsum(dO o O):
//   dV = P^T dO      dP = dO V^T      dS = P o (dP - delta) * scale      dQ = dS K      dK = dS^T Q
// P is recomputed from Q, K and the forward's log-sum-exp (no (L x L) tensor is ever stored).  Two kernels
// with the forward's operand maps, so every product keeps "one lane = one column":
//   attn_bwd_dq : lane = query.  S^T = K.Q^T and dP^T = V.dO^T (A from LDS rows, B = register fragments of Q
//                 and dO), then dQ^T[d][q] += K^T[d][key] . dS^T[key][q] (A = LDS image of the pre-transposed
//                 K, B = dS straight from the accumulator registers).
//   attn_bwd_dkv: lane = key.    S = Q.K^T and dP = dO.V^T (A = Q / dO rows from LDS, B = register fragments
//                 of K and V), then dV^T[d][key] += dO^T[d][q] . P[q][key] and dK^T[d][key] += Q^T[d][q] . dS[q][key]
//                 (A = LDS images of the pre-transposed dO and Q).
// =====================================================================================================

template <bool PLAIN = false>
__device__ __forceinline__ void attn_bwd_dq_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                 const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                 const float *__restrict__ LSE, const __bf16 *__restrict__ O,
                                                 float *__restrict__ DELTA, __bf16 *__restrict__ dQ, const BwdDims &dm,
                                                 const int bx, const int bh) {
  __shared__ __align__(16) unsigned char s_k[2][AT_KB * 128];  // double-buffered: see attn_fwd_kernel
  __shared__ __align__(16) unsigned char s_v[2][AT_KB * 128];
  const float scale = dm.scale;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int b = bh / dm.H, hd = bh % dm.H;
  const unsigned seed = eff_seed(dm);
  const int q0 = bx * AT_QB + wid * AT_QW;
  const int live = attn_live_waves(dm.Lq, bx * AT_QB);
  if (wid >= live) return;
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs;
  const __bf16 *Gb = dO + b * dm.o_bs + hd * dm.o_hs;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs;
  const float *mrow = dm.mask ? dm.mask + (long)b * dm.Lkp : nullptr;
  const float c = scale * 1.4426950408889634f;

  bf16x8 qf[4], gf[4];
  const int qr = min(q0 + r, dm.Lq - 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8 *>(Qb + (long)qr * dm.q_rs + 16 * s + 8 * h);
    gf[s] = *reinterpret_cast<const bf16x8 *>(Gb + (long)qr * dm.o_rs + 16 * s + 8 * h);
  }
  const float lse = LSE[(long)bh * dm.Lq + qr];
  // delta[q] = rowsum(dO o O), computed here (O is the forward's contiguous (B, Lq, H, 64) output) and published
  // for the dK/dV kernel that follows on the same stream
  float delta = 0.0f;
  {
    const __bf16 *Orow = O + (((long)b * dm.Lq + qr) * dm.H + hd) * AT_D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(Orow + 16 * s + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)ov[j] * (float)gf[s][j];
    }
    delta = xhalf_sum(delta);
    if (h == 0 && q0 + r < dm.Lq) DELTA[(long)bh * dm.Lq + q0 + r] = delta;
  }
  f32x16 a0 = {0}, a1 = {0};
  const int nkt = (dm.Lk + AT_KB - 1) / AT_KB;
  TileDma<true> tdma;   // K / V tiles by LDS-DMA (see attn_fwd_kernel)
  tdma.init(Kb, dm.k_rs, Vb, dm.k_rs, dm.Lk, lane, wid, live);
  auto turn = [&](int kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) tdma.issue(kt + 1, s_k[(kt + 1) & 1], s_v[(kt + 1) & 1]);
  };
  tdma.issue(0, s_k[0], s_v[0]);
  const bool active = q0 < dm.Lq;  // wave-uniform
  if (!PLAIN) {
    for (int kt = 0; kt < nkt; ++kt) {
      turn(kt);
      if (active)
        dq_tile(s_k[kt & 1], s_v[kt & 1], qf, gf, dm, mrow, c, scale, lse, delta, seed, bh, q0 + r, kt, kt == nkt - 1, r, h, a0, a1);
    }
  } else {
    for (int kt = 0; kt < nkt - 1; ++kt) {
      turn(kt);
      if (active)
        dq_tile<true, false>(s_k[kt & 1], s_v[kt & 1], qf, gf, dm, nullptr, c, scale, lse, delta, seed, bh, q0 + r, kt, false, r, h, a0, a1);
    }
    turn(nkt - 1);
    if (active)
      dq_tile<true, true>(s_k[(nkt - 1) & 1], s_v[(nkt - 1) & 1], qf, gf, dm, nullptr, c, scale, lse, delta, seed, bh, q0 + r, nkt - 1, true, r, h, a0, a1);
  }
  if (q0 + r < dm.Lq) store_T(dQ + b * dm.q_bs + hd * dm.q_hs + (long)(q0 + r) * dm.q_rs, a0, a1, h, scale);
}

template <int MINW, bool PLAIN = false>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dq_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                          const __bf16 *__restrict__ V,
                                                          const __bf16 *__restrict__ dO, const float *__restrict__ LSE,
                                                          const __bf16 *__restrict__ O, float *__restrict__ DELTA,
                                                          __bf16 *__restrict__ dQ, BwdDims dm) {
  attn_bwd_dq_body<PLAIN>(Q, K, V, dO, LSE, O, DELTA, dQ, dm, blockIdx.x, blockIdx.y);
}

// Lq <= 32: the four waves share the queries and split the key tiles (see attn_fwd_narrow_kernel); dQ^T partials are
// summed through LDS in wave order.
__device__ __forceinline__ void attn_bwd_dq_narrow_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                        const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                        const float *__restrict__ LSE, const __bf16 *__restrict__ O,
                                                        float *__restrict__ DELTA, __bf16 *__restrict__ dQ,
                                                        const BwdDims &dm, const int bh) {
  // (LDS: the merge buffer aliases the tile images, as in attn_fwd_narrow_body: two workgroups per CU)
  __shared__ __align__(16) unsigned char s_kv[2 * AT_NW * AT_KB * 128];
  unsigned char (*s_k)[AT_KB * 128] = reinterpret_cast<unsigned char (*)[AT_KB * 128]>(s_kv);
  unsigned char (*s_v)[AT_KB * 128] = reinterpret_cast<unsigned char (*)[AT_KB * 128]>(s_kv + AT_NW * AT_KB * 128);
  float (*s_o)[AT_D][33] = reinterpret_cast<float (*)[AT_D][33]>(s_kv);
  const float scale = dm.scale;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int b = bh / dm.H, hd = bh % dm.H;
  const unsigned seed = eff_seed(dm);
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs;
  const __bf16 *Gb = dO + b * dm.o_bs + hd * dm.o_hs;
  const __bf16 *Kb = K + b * dm.k_bs + hd * dm.k_hs;
  const __bf16 *Vb = V + b * dm.k_bs + hd * dm.k_hs;
  const float *mrow = dm.mask ? dm.mask + (long)b * dm.Lkp : nullptr;
  const float c = scale * 1.4426950408889634f;
  bf16x8 qf[4], gf[4];
  const int qr = min(r, dm.Lq - 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8 *>(Qb + (long)qr * dm.q_rs + 16 * s + 8 * h);
    gf[s] = *reinterpret_cast<const bf16x8 *>(Gb + (long)qr * dm.o_rs + 16 * s + 8 * h);
  }
  const float lse = LSE[(long)bh * dm.Lq + qr];
  float delta = 0.0f;
  {
    const __bf16 *Orow = O + (((long)b * dm.Lq + qr) * dm.H + hd) * AT_D;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ov = *reinterpret_cast<const bf16x8 *>(Orow + 16 * s + 8 * h);
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += (float)ov[j] * (float)gf[s][j];
    }
    delta = xhalf_sum(delta);
    if (wid == 0 && h == 0 && r < dm.Lq) DELTA[(long)bh * dm.Lq + r] = delta;
  }
  f32x16 a0 = {0}, a1 = {0};
  const __bf16 *K2b = dm.K2 ? dm.K2 + b * dm.k2_bs + hd * dm.k2_hs : nullptr;  // second key/value segment (AttnDims)
  const __bf16 *V2b = dm.V2 ? dm.V2 + b * dm.k2_bs + hd * dm.k2_hs : nullptr;
  const int nkt = dm.nkt1 + (dm.Lk2 + AT_KB - 1) / AT_KB;
  {  // (per-wave pipeline without workgroup barriers, next tile's loads in flight: see attn_fwd_narrow_body)
    uint4 kr[8], vr[8];
    int kt = wid;
    KeyTile nx_ = key_tile(dm, Kb, Vb, K2b, V2b, min(kt, nkt - 1));
    if (kt < nkt) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kr[j] = stage_load(nx_.k, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
        vr[j] = stage_load(nx_.v, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
      }
    }
    for (; kt < nkt; kt += AT_NW) {
      const KeyTile kt_ = nx_;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        stage_store(s_k[wid], lane + 64 * j, kr[j]);
        stage_store(s_v[wid], lane + 64 * j, vr[j]);
      }
      if (kt + AT_NW < nkt) {
        nx_ = key_tile(dm, Kb, Vb, K2b, V2b, kt + AT_NW);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kr[j] = stage_load(nx_.k, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
          vr[j] = stage_load(nx_.v, nx_.rs, nx_.row0, nx_.nrows, lane + 64 * j);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      dq_tile(s_k[wid], s_v[wid], qf, gf, dm, mrow, c, scale, lse, delta, seed, bh, r, kt, kt_.last, r, h, a0, a1,
              kt_.kend);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();   // every wave has finished with its images: the merge buffer aliases them
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s_o[wid][crow(i, h)][r] = a0[i];
    s_o[wid][32 + crow(i, h)][r] = a1[i];
  }
  __syncthreads();
  const int q = t & 31, d0 = (t >> 5) * 8;
  if (q < dm.Lq) {
    bf16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      out[j] = (__bf16)(((s_o[0][d0 + j][q] + s_o[1][d0 + j][q]) + (s_o[2][d0 + j][q] + s_o[3][d0 + j][q])) * scale);
    *reinterpret_cast<bf16x8 *>(dQ + b * dm.q_bs + hd * dm.q_hs + (long)q * dm.q_rs + d0) = out;
  }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_narrow_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                                 const __bf16 *__restrict__ V,
                                                                 const __bf16 *__restrict__ dO, const float *__restrict__ LSE,
                                                                 const __bf16 *__restrict__ O, float *__restrict__ DELTA,
                                                                 __bf16 *__restrict__ dQ, BwdDims dm) {
  attn_bwd_dq_narrow_body(Q, K, V, dO, LSE, O, DELTA, dQ, dm, blockIdx.y);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_narrow_pair_kernel(const AttnPair a) {
  const int g = blockIdx.z;
  attn_bwd_dq_narrow_body(a.Q[g], a.K[g], a.V[g], a.dO[g], a.LSE[g], a.O[g], a.DELTA[g], a.out[g], a.dm[g], blockIdx.y);
}

// IDELTA: delta[q] = rowsum(dO o O) is computed here from O (contiguous (B, Lq, H, 64)) instead of read from DELTA -- for
// the launch that runs the dQ and the dK/dV pass of a SMALL attention side by side (attn_bwd_small_kernel), where the
// dQ pass's DELTA is not ordered before this pass
template <bool PLAIN = false, bool IDELTA = false>
__device__ __forceinline__ void attn_bwd_dkv_body(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                  const __bf16 *__restrict__ V, const __bf16 *__restrict__ dO,
                                                  const float *__restrict__ LSE, const float *__restrict__ DELTA,
                                                  __bf16 *__restrict__ dK, __bf16 *__restrict__ dV, const BwdDims &dm,
                                                  const int bx, const int bh, const __bf16 *__restrict__ O = nullptr) {
  __shared__ __align__(16) unsigned char s_q[2][AT_KB * 128];  // double-buffered: see attn_fwd_kernel
  __shared__ __align__(16) unsigned char s_g[2][AT_KB * 128];
  __shared__ __align__(16) float s_lse[2][AT_KB];
  __shared__ __align__(16) float s_del[2][AT_KB];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, r = lane & 31, h = lane >> 5;
  const int b = bh / dm.H, hd = bh % dm.H;
  const unsigned seed = eff_seed(dm);
  // key blocks of segment 1 first, then those of the optional second segment (AttnDims): a block never straddles
  const int nb1 = (dm.Lk + AT_QB - 1) / AT_QB;
  const bool seg2 = !PLAIN && bx >= nb1;   // wave-uniform (PLAIN is never used with two segments)
  const int k0 = (seg2 ? bx - nb1 : bx) * AT_QB + wid * AT_QW;  // first key of this wave, in its segment
  const int Lks = seg2 ? dm.Lk2 : dm.Lk;               // keys in this block's segment
  const int live = attn_live_waves(Lks, (seg2 ? bx - nb1 : bx) * AT_QB);
  if (wid >= live) return;
  const int kpad0 = seg2 ? dm.nkt1 * 64 : 0;           // padded key index (mask row, dropout hash) of the segment's key 0
  const long ks_rs = seg2 ? dm.k2_rs : dm.k_rs;
  const long ks_off = seg2 ? b * dm.k2_bs + hd * dm.k2_hs : b * dm.k_bs + hd * dm.k_hs;
  const __bf16 *Qb = Q + b * dm.q_bs + hd * dm.q_hs;
  const __bf16 *Gb = dO + b * dm.o_bs + hd * dm.o_hs;
  const __bf16 *Kb = (seg2 ? dm.K2 : K) + ks_off;
  const __bf16 *Vb = (seg2 ? dm.V2 : V) + ks_off;
  const float *lseb = LSE + (long)bh * dm.Lq, *delb = DELTA + (long)bh * dm.Lq;
  const float scale = dm.scale;
  const float c = scale * 1.4426950408889634f;

  bf16x8 kf[4], vf[4];
  const int kr = min(k0 + r, Lks - 1);
  const float mkey = (!PLAIN && dm.mask) ? dm.mask[(long)b * dm.Lkp + kpad0 + kr] : 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = *reinterpret_cast<const bf16x8 *>(Kb + (long)kr * ks_rs + 16 * s + 8 * h);
    vf[s] = *reinterpret_cast<const bf16x8 *>(Vb + (long)kr * ks_rs + 16 * s + 8 * h);
  }
  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  const int nqt = (dm.Lq + AT_KB - 1) / AT_KB;
  // Q / dO tiles by LDS-DMA (TileDma); the 64 log-sum-exp / delta values of a tile still travel through two registers of
  // the first wave and are written to LDS right after the barrier, BEFORE the next tile's DMAs are issued (hipcc fences a
  // ds_write against LDS-DMAs in flight with vmcnt(0))
  TileDma<false> tdma;
  tdma.init(Qb, dm.q_rs, Gb, dm.o_rs, dm.Lq, lane, wid, live);
  float rl = 0.f, rd = 0.f;
  auto fetch = [&](int qt) {   // the row scalars of tile qt into registers
    if (t < AT_KB) {
      const int qq = min(qt * AT_KB + t, dm.Lq - 1);
      rl = lseb[qq];
      if (IDELTA) {
        const __bf16 *grow = Gb + (long)qq * dm.o_rs, *orow = O + (((long)b * dm.Lq + qq) * dm.H + hd) * AT_D;
        float acc = 0.0f;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const bf16x8 gv = *reinterpret_cast<const bf16x8 *>(grow + 8 * s8), ov = *reinterpret_cast<const bf16x8 *>(orow + 8 * s8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc += (float)ov[j] * (float)gv[j];
        }
        rd = acc;
      } else {
        rd = delb[qq];
      }
    }
  };
  auto commit = [&](int buf) {
    if (t < AT_KB) { s_lse[buf][t] = rl; s_del[buf][t] = rd; }
  };
  // before tile qt is consumed: own DMAs / scalar loads landed, barrier (all images visible, tile qt - 1's no longer read),
  // then tile qt + 1: scalars committed, DMAs issued, the scalars of tile qt + 2 requested
  auto turn = [&](int qt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qt + 1 < nqt) {
      commit((qt + 1) & 1);
      tdma.issue(qt + 1, s_q[(qt + 1) & 1], s_g[(qt + 1) & 1]);
      if (qt + 2 < nqt) fetch(qt + 2);
    }
  };
  fetch(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  commit(0);
  tdma.issue(0, s_q[0], s_g[0]);
  if (nqt > 1) fetch(1);
  const bool active = k0 < Lks;  // wave-uniform: a wave whose 32 keys are all past the end only helps staging
  if (!PLAIN) {
    for (int qt = 0; qt < nqt; ++qt) {
      turn(qt);
      if (active)
        dkv_tile(s_q[qt & 1], s_g[qt & 1], s_lse[qt & 1], s_del[qt & 1], kf, vf, dm, c, mkey, seed, bh, kpad0 + k0 + r, qt,
                 qt == nqt - 1, r, h, dk0, dk1, dv0, dv1);
    }
  } else {
    for (int qt = 0; qt < nqt - 1; ++qt) {
      turn(qt);
      if (active)
        dkv_tile<true, false>(s_q[qt & 1], s_g[qt & 1], s_lse[qt & 1], s_del[qt & 1], kf, vf, dm, c, 0.0f, seed, bh,
                              k0 + r, qt, false, r, h, dk0, dk1, dv0, dv1);
    }
    turn(nqt - 1);
    const int lb = (nqt - 1) & 1;
    if (active)
      dkv_tile<true, true>(s_q[lb], s_g[lb], s_lse[lb], s_del[lb], kf, vf, dm, c, 0.0f, seed, bh, k0 + r, nqt - 1, true,
                           r, h, dk0, dk1, dv0, dv1);
  }
  if (k0 + r < Lks) {
    const long off = ks_off + (long)(k0 + r) * ks_rs;
    store_T((seg2 ? dm.dK2 : dK) + off, dk0, dk1, h, scale);
    store_T((seg2 ? dm.dV2 : dV) + off, dv0, dv1, h, 1.0f);
  }
}

template <int MINW, bool PLAIN = false>
__global__ __launch_bounds__(256, MINW) void attn_bwd_dkv_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                           const __bf16 *__restrict__ V,
                                                           const __bf16 *__restrict__ dO,
                                                           const float *__restrict__ LSE, const float *__restrict__ DELTA,
                                                           __bf16 *__restrict__ dK, __bf16 *__restrict__ dV, BwdDims dm) {
  attn_bwd_dkv_body<PLAIN>(Q, K, V, dO, LSE, DELTA, dK, dV, dm, blockIdx.x, blockIdx.y);
}

// the dK / dV passes of an AttnPair: the grid's x extent is the larger of the two key-block counts
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_pair_kernel(const AttnPair a) {
  const int g = blockIdx.z;
  const BwdDims &dm = a.dm[g];
  const int nb = (dm.Lk + AT_QB - 1) / AT_QB + (dm.Lk2 > 0 ? (dm.Lk2 + AT_QB - 1) / AT_QB : 0);
  if ((int)blockIdx.x >= nb) return;
  attn_bwd_dkv_body<false>(a.Q[g], a.K[g], a.V[g], a.dO[g], a.LSE[g], a.DELTA[g], a.dK[g], a.dV[g], dm, blockIdx.x, blockIdx.y);
}

// A SMALL attention's backward (at most one 128-row block of queries and of keys: the 20-token self-attentions of the twin
// levels, the 5-token decoder's self- and cross-attentions) in ONE launch: blockIdx.z = 0 runs the dQ pass, 1 the dK/dV pass
// with its own delta -- both are ~6 us launches at the floor of what a launch costs, 36 of each per c3 step.
__global__ __launch_bounds__(256, 2) void attn_bwd_small_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                                const __bf16 *__restrict__ V,
                                                                const __bf16 *__restrict__ dO,
                                                                const float *__restrict__ LSE,
                                                                const __bf16 *__restrict__ O, float *__restrict__ DELTA,
                                                                __bf16 *__restrict__ dQ, __bf16 *__restrict__ dK,
                                                                __bf16 *__restrict__ dV, BwdDims dm) {
  if (blockIdx.z == 0) attn_bwd_dq_body<false>(Q, K, V, dO, LSE, O, DELTA, dQ, dm, 0, blockIdx.y);
  else attn_bwd_dkv_body<false, true>(Q, K, V, dO, LSE, DELTA, dK, dV, dm, 0, blockIdx.y, O);
}

}  // namespace bq

using namespace bq;

// persistent-grid switch (measurement: tools/bench_attn.py times both); bit 0: forward, bit 1: dQ pass, bit 2: dK/dV pass
// Default 0 (round 6): alone the resident grid is ahead at the ViT shape (forward 84.8 us against 88.5, backward 233.0 against
// 235.2, tools/ab_attn.py, one box), inside the c3 step it is behind (34.0-34.1 ms against 33.8, tools/ab_step_attn.sh):
// its static item walk cannot give way to the detector stream's workgroups the way block-by-block dispatch does.
static int g_attn_persist = 0;
extern "C" __attribute__((visibility("default"))) int bq_attn_set_persistent(int mask) {
  const int prev = g_attn_persist;
  g_attn_persist = mask;
  return prev;
}
static int attn_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount;
  }
  return n;
}

// Q: bf16 (B, Lq, H, 64) / K, V: (B, Lk, H, 64) given by element strides (batch, token, head), 64 contiguous
// elements per (token, head); V is strided like K.  (No transposed copies: the kernels read V^T / K^T / Q^T / dO^T out of
// the row-major LDS tiles with ds_read_b64_tr_b16.)  Lkp = row length of the mask, a multiple of 64 >= Lk.
// O: bf16 strided like Q's shape; LSE: f32 [B*H][Lq], log2-domain log-sum-exp of the scaled (+masked) scores.
// mask: optional f32 [B][Lkp] additive key mask ALREADY multiplied by log2(e) (0 in the padding); p_drop / seed:
// dropout on the attention probabilities (stateless hash, regenerated by the backward); the effective seed is
// seed_ptr[0] * 2654435761 + seed when seed_ptr (a device counter the caller bumps once per step) is given.
// causal != 0 (Lq == Lk): keys after the query are masked as well.
extern "C" __attribute__((visibility("default"))) int bq_attn_fwd(
    const void *Q, const void *K, const void *V, void *O, float *LSE, const float *mask, int B, int H, int Lq, int Lk,
    int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long o_bs, long o_rs, long o_hs,
    float scale, float p_drop, unsigned seed, const unsigned *seed_ptr, int causal, void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && Lkp >= Lk && Lkp % 64 == 0, BQ_EINVAL, "attn_fwd: bad extents");
  BQ_REQUIRE(!causal || Lq == Lk, BQ_EINVAL, "attn_fwd: causal needs Lq == Lk");
  BQ_REQUIRE(Q && K && V && O && LSE, BQ_EINVAL, "attn_fwd: null pointer");
  BQ_REQUIRE((q_rs % 8) == 0 && (k_rs % 8) == 0 && (o_rs % 4) == 0 && (q_hs % 8) == 0 && (k_hs % 8) == 0, BQ_EINVAL,
             "attn_fwd: rows must be 16-byte aligned");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_fwd: bad dropout probability");
  BQ_REQUIRE(q_rs > 0 && k_rs > 0 && q_rs < (1 << 23) && k_rs < (1 << 23), BQ_EINVAL, "attn_fwd: row stride out of range");
  AttnDims dm{B, H, Lq, Lk, 0, Lkp, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, o_bs, o_rs, o_hs, mask, scale,
              1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr, causal ? 1 : 0};
  dm.nkt1 = (Lk + AT_KB - 1) / AT_KB;  // single key/value segment
  if (Lq <= AT_QW && Lk > 2 * AT_KB && (o_rs % 8) == 0 && (o_hs % 8) == 0) {
    hipLaunchKernelGGL(attn_fwd_narrow_kernel, dim3(1, B * H), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Q,
                       (const __bf16 *)K, (const __bf16 *)V, (__bf16 *)O, LSE, dm);
    return check_launch("attn_fwd_narrow");
  }
  const bool plain = !mask && !causal && dm.drop_thresh == 0;
  if (plain && (g_attn_persist & 1)) {
    // resident grid (attn_fwd_persist_kernel): 3 workgroups per CU, every one walks (head, query block) items
    const int rag = Lq % AT_QB, nqb = Lq / AT_QB + (rag > 1 ? 1 : 0), slots = 3 * attn_cu_count();
    // (few rounds only: with many items per workgroup the dispatcher's dynamic order balances better than a static walk --
    // L = 4097: 2.33 ms resident against 2.19 ms block by block; four workgroups per CU, i.e. 1.5 items each: 90.8 us
    // against 86.5 at L = 1025)
    if ((B * H) % 8 == 0 && (long)B * H * nqb >= slots && (long)B * H * nqb <= 4L * slots && slots % 8 == 0) {
      return attn_fwd_persist_launch(Q, K, V, O, LSE, dm, nqb, (rag == 1 ? 1 : 0) | ((g_attn_persist & 8) ? 2 : 0), slots, (hipStream_t)stream);
    }
  }
  const dim3 grid((Lq + AT_QB - 1) / AT_QB, B * H);
  // 3 waves / SIMD, and for the ViT's case (no mask, not causal, no dropout) the PLAIN instantiation with the last tile
  // peeled: tools/bench_attn.py sweeps of round 1 / 2 (2 waves: slower; EARLY score issue: no gain)
  if (plain)
    hipLaunchKernelGGL((attn_fwd_kernel<BQ_ATTN_FWD_MINW, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (__bf16 *)O, LSE, dm);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<3, 0>), grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (__bf16 *)O, LSE, dm);
  return check_launch("attn_fwd");
}

// Backward of bq_attn_fwd.  dQ shares Q's strides, dK/dV share K's (V must be strided like K), dO has its own.
// LSE and O (contiguous (B,Lq,H,64)) from the forward; DELTA: f32 [B*H][Lq] scratch (rowsum(dO*O), produced by the dQ
// kernel, consumed by the dK/dV kernel).  mask / p_drop / seed / causal exactly as given to the forward.
extern "C" __attribute__((visibility("default"))) int bq_attn_bwd(
    const void *Q, const void *K, const void *V, const void *dO, const float *LSE, const void *O, float *DELTA,
    const float *mask, void *dQ, void *dK, void *dV, int B, int H, int Lq, int Lk, int Lkp, long q_bs, long q_rs,
    long q_hs, long k_bs, long k_rs, long k_hs, long g_bs, long g_rs, long g_hs, float scale, float p_drop, unsigned seed,
    const unsigned *seed_ptr, int causal, void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && Lkp >= Lk && Lkp % 64 == 0, BQ_EINVAL, "attn_bwd: bad extents");
  BQ_REQUIRE(!causal || Lq == Lk, BQ_EINVAL, "attn_bwd: causal needs Lq == Lk");
  BQ_REQUIRE(Q && K && V && dO && LSE && O && DELTA && dQ && dK && dV, BQ_EINVAL, "attn_bwd: null pointer");
  BQ_REQUIRE((q_rs % 8) == 0 && (k_rs % 8) == 0 && (g_rs % 8) == 0 && (q_hs % 8) == 0 && (k_hs % 8) == 0 &&
                 (g_hs % 8) == 0, BQ_EINVAL, "attn_bwd: rows must be 16-byte aligned");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_bwd: bad dropout probability");
  BQ_REQUIRE(q_rs > 0 && k_rs > 0 && g_rs > 0 && q_rs < (1 << 23) && k_rs < (1 << 23) && g_rs < (1 << 23), BQ_EINVAL,
             "attn_bwd: row stride out of range");
  BwdDims dm{B, H, Lq, Lk, 0, Lkp, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, g_bs, g_rs, g_hs, mask, scale,
             1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr, causal ? 1 : 0};
  dm.nkt1 = (Lk + AT_KB - 1) / AT_KB;  // single key/value segment
  hipStream_t st = (hipStream_t)stream;
  constexpr int dq_w = BQ_ATTN_DQ_MINW, dkv_w = 2;   // waves / SIMD: measured (dK/dV at 2: 0.34 -> 0.25 ms, round 1)
  const bool plain = !mask && !causal && dm.drop_thresh == 0;
#define BQ_DQ(W, P) hipLaunchKernelGGL((attn_bwd_dq_kernel<W, P>), dim3((Lq + AT_QB - 1) / AT_QB, B * H), dim3(256), 0, \
                                       st, (const __bf16 *)Q, (const __bf16 *)K, (const __bf16 *)V, (const __bf16 *)dO, \
                                       LSE, (const __bf16 *)O, DELTA, (__bf16 *)dQ, dm)
  if (!plain && Lq <= AT_QB && Lk <= AT_QB) {
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(1, B * H, 2), dim3(256), 0, st, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (const __bf16 *)dO, LSE, (const __bf16 *)O, DELTA, (__bf16 *)dQ, (__bf16 *)dK,
                       (__bf16 *)dV, dm);
    return check_launch("attn_bwd_small");
  }
  const int rag_q = Lq % AT_QB, nqb_p = Lq / AT_QB + (rag_q > 1 ? 1 : 0), slots3 = 3 * attn_cu_count();
  if (plain && (g_attn_persist & 2) && (B * H) % 8 == 0 && slots3 % 8 == 0 && (long)B * H * nqb_p >= slots3 &&
      (long)B * H * nqb_p <= 4L * slots3)
    attn_bwd_dq_persist_launch(Q, K, V, dO, LSE, O, DELTA, dQ, dm, nqb_p, rag_q == 1 ? 1 : 0, slots3, st);   // resident grid, 3 per CU
  else if (Lq <= AT_QW && Lk > 2 * AT_KB)
    hipLaunchKernelGGL(attn_bwd_dq_narrow_kernel, dim3(1, B * H), dim3(256), 0, st, (const __bf16 *)Q, (const __bf16 *)K,
                       (const __bf16 *)V, (const __bf16 *)dO, LSE, (const __bf16 *)O, DELTA, (__bf16 *)dQ, dm);
  else
  {
    if (plain) BQ_DQ(dq_w, true); else BQ_DQ(2, false);   // (the general instantiation would spill 15 registers at four per CU)
  }
#undef BQ_DQ
  int rc = check_launch("attn_bwd_dq");
  if (rc) return rc;
#define BQ_DKV(W, P) hipLaunchKernelGGL((attn_bwd_dkv_kernel<W, P>), dim3((Lk + AT_QB - 1) / AT_QB, B * H), dim3(256), 0, \
                                        st, (const __bf16 *)Q, (const __bf16 *)K, (const __bf16 *)V, (const __bf16 *)dO,  \
                                        LSE, DELTA, (__bf16 *)dK, (__bf16 *)dV, dm)
  const int rag_k = Lk % AT_QB, nkb_p = Lk / AT_QB + (rag_k > 1 ? 1 : 0), slots2 = 2 * attn_cu_count();
  if (plain && (g_attn_persist & 4) && (B * H) % 8 == 0 && slots2 % 8 == 0 && Lq > AT_KB && (long)B * H * nkb_p >= slots2 &&
      (long)B * H * nkb_p <= 6L * slots2)
    attn_bwd_dkv_persist_launch(Q, K, V, dO, LSE, DELTA, dK, dV, dm, nkb_p, rag_k == 1 ? 1 : 0, slots2, st);   // resident grid, 2 per CU
  else if (plain) BQ_DKV(dkv_w, true); else BQ_DKV(dkv_w, false);
#undef BQ_DKV
  return check_launch("attn_bwd_dkv");
}


// ---- two key/value segments --------------------------------------------------------------------------------
// Attention of Lq <= 32 queries over cat(segment 1, segment 2) along the key axis WITHOUT the concatenated
// tensor: the twin cross-attention of the reference (med.py:549-562) attends to cat(image tokens, other stream's
// text states); with two segments the image-token K/V of all layers can come from one hoisted projection and the
// per-layer 25 MB concatenation (and the strided slicing of its gradient) disappears.
// K / V: segment 1, (B, Lk, H, 64) by strides k_*; K2 / V2: segment 2, (B, Lk2, H, 64) by strides k2_*.
// mask: optional f32 [B][Lkp], Lkp = 64 * (ceil(Lk / 64) + ceil(Lk2 / 64)): segment 1's keys at [0, Lk), segment
// 2's at [64 * ceil(Lk / 64), ... + Lk2), already multiplied by log2(e), 0 in the padding.  Everything else as
// bq_attn_fwd (the kernels are the narrow-query ones: the four waves split the 64-key tiles of both segments).
extern "C" __attribute__((visibility("default"))) int bq_attn_fwd2(
    const void *Q, const void *K, const void *V, const void *K2, const void *V2, void *O, float *LSE, const float *mask,
    int B, int H, int Lq, int Lk, int Lk2, int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs,
    long k2_bs, long k2_rs, long k2_hs, long o_bs, long o_rs, long o_hs, float scale, float p_drop, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lq <= AT_QW && Lk > 0 && Lk2 > 0, BQ_EINVAL,
             "attn_fwd2: bad extents (needs 1 <= Lq <= 32 and two non-empty segments)");
  const int nkt1 = (Lk + AT_KB - 1) / AT_KB, nkt2 = (Lk2 + AT_KB - 1) / AT_KB;
  BQ_REQUIRE(Lkp == 64 * (nkt1 + nkt2), BQ_EINVAL, "attn_fwd2: mask row length must be 64 * (tiles of both segments)");
  BQ_REQUIRE(Q && K && V && K2 && V2 && O && LSE, BQ_EINVAL, "attn_fwd2: null pointer");
  BQ_REQUIRE((q_rs % 8) == 0 && (k_rs % 8) == 0 && (k2_rs % 8) == 0 && (o_rs % 8) == 0 && (q_hs % 8) == 0 &&
                 (k_hs % 8) == 0 && (k2_hs % 8) == 0 && (o_hs % 8) == 0, BQ_EINVAL,
             "attn_fwd2: rows must be 16-byte aligned");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_fwd2: bad dropout probability");
  AttnDims dm{B, H, Lq, Lk, 0, Lkp, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, o_bs, o_rs, o_hs, mask, scale,
              1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr, 0};
  dm.Lk2 = Lk2; dm.nkt1 = nkt1;
  dm.k2_bs = k2_bs; dm.k2_rs = k2_rs; dm.k2_hs = k2_hs;
  dm.K2 = (const __bf16 *)K2; dm.V2 = (const __bf16 *)V2;
  hipLaunchKernelGGL(attn_fwd_narrow_kernel, dim3(1, B * H), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)Q,
                     (const __bf16 *)K, (const __bf16 *)V, (__bf16 *)O, LSE, dm);
  return check_launch("attn_fwd2");
}

// Backward of bq_attn_fwd2: dQ like Q; dK / dV strided like K (segment 1), dK2 / dV2 like K2 (segment 2).
extern "C" __attribute__((visibility("default"))) int bq_attn_bwd2(
    const void *Q, const void *K, const void *V, const void *K2, const void *V2, const void *dO, const float *LSE,
    const void *O, float *DELTA, const float *mask, void *dQ, void *dK, void *dV, void *dK2, void *dV2, int B, int H,
    int Lq, int Lk, int Lk2, int Lkp, long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, long k2_bs,
    long k2_rs, long k2_hs, long g_bs, long g_rs, long g_hs, float scale, float p_drop, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lq <= AT_QW && Lk > 0 && Lk2 > 0, BQ_EINVAL,
             "attn_bwd2: bad extents (needs 1 <= Lq <= 32 and two non-empty segments)");
  const int nkt1 = (Lk + AT_KB - 1) / AT_KB, nkt2 = (Lk2 + AT_KB - 1) / AT_KB;
  BQ_REQUIRE(Lkp == 64 * (nkt1 + nkt2), BQ_EINVAL, "attn_bwd2: mask row length must be 64 * (tiles of both segments)");
  BQ_REQUIRE(Q && K && V && K2 && V2 && dO && LSE && O && DELTA && dQ && dK && dV && dK2 && dV2, BQ_EINVAL,
             "attn_bwd2: null pointer");
  BQ_REQUIRE((q_rs % 8) == 0 && (k_rs % 8) == 0 && (k2_rs % 8) == 0 && (g_rs % 8) == 0 && (q_hs % 8) == 0 &&
                 (k_hs % 8) == 0 && (k2_hs % 8) == 0 && (g_hs % 8) == 0, BQ_EINVAL,
             "attn_bwd2: rows must be 16-byte aligned");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_bwd2: bad dropout probability");
  BQ_REQUIRE(q_rs < (1 << 23) && g_rs < (1 << 23), BQ_EINVAL, "attn_bwd2: row stride out of range");
  BwdDims dm{B, H, Lq, Lk, 0, Lkp, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, g_bs, g_rs, g_hs, mask, scale,
             1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr, 0};
  dm.Lk2 = Lk2; dm.nkt1 = nkt1;
  dm.k2_bs = k2_bs; dm.k2_rs = k2_rs; dm.k2_hs = k2_hs;
  dm.K2 = (const __bf16 *)K2; dm.V2 = (const __bf16 *)V2;
  dm.dK2 = (__bf16 *)dK2; dm.dV2 = (__bf16 *)dV2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_narrow_kernel, dim3(1, B * H), dim3(256), 0, st, (const __bf16 *)Q, (const __bf16 *)K,
                     (const __bf16 *)V, (const __bf16 *)dO, LSE, (const __bf16 *)O, DELTA, (__bf16 *)dQ, dm);
  int rc = check_launch("attn_bwd2_dq");
  if (rc) return rc;
  const int nb = (Lk + AT_QB - 1) / AT_QB + (Lk2 + AT_QB - 1) / AT_QB;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, false>), dim3(nb, B * H), dim3(256), 0, st, (const __bf16 *)Q,
                     (const __bf16 *)K, (const __bf16 *)V, (const __bf16 *)dO, LSE, DELTA, (__bf16 *)dK, (__bf16 *)dV, dm);
  return check_launch("attn_bwd2_dkv");
}

// ---- two narrow attentions per launch (AttnPair) ----------------------------------------------------------------------
static int fill_pair(const bq_attn_side *sd, int B, int H, float scale, float p_drop, const unsigned *seed_ptr, bool backward,
                     AttnPair &a) {
  for (int g = 0; g < 2; ++g) {
    const bq_attn_side &s = sd[g];
    BQ_REQUIRE(s.Lq > 0 && s.Lq <= AT_QW && s.Lk > 2 * AT_KB && s.Lkp >= s.Lk && s.Lkp % 64 == 0, BQ_EINVAL,
               "attn pair: side %d needs 1 <= Lq <= 32 queries and more than 128 keys (Lq %d, Lk %d)", g, s.Lq, s.Lk);
    BQ_REQUIRE(s.Q && s.K && s.V && s.out && s.LSE, BQ_EINVAL, "attn pair: null pointer (side %d)", g);
    BQ_REQUIRE(!backward || (s.dO && s.O && s.DELTA && s.dK && s.dV), BQ_EINVAL, "attn pair: null pointer (side %d)", g);
    BQ_REQUIRE((s.q_rs % 8) == 0 && (s.k_rs % 8) == 0 && (s.o_rs % 8) == 0 && (s.q_hs % 8) == 0 && (s.k_hs % 8) == 0 &&
                   (s.o_hs % 8) == 0, BQ_EINVAL, "attn pair: rows must be 16-byte aligned");
    BQ_REQUIRE(s.q_rs > 0 && s.k_rs > 0 && s.q_rs < (1 << 23) && s.k_rs < (1 << 23), BQ_EINVAL,
               "attn pair: row stride out of range");
    a.Q[g] = (const __bf16 *)s.Q; a.K[g] = (const __bf16 *)s.K; a.V[g] = (const __bf16 *)s.V;
    a.dO[g] = (const __bf16 *)s.dO; a.O[g] = (const __bf16 *)s.O;
    a.LSE[g] = s.LSE; a.LSEw[g] = s.LSE; a.DELTA[g] = s.DELTA;
    a.out[g] = (__bf16 *)s.out; a.dK[g] = (__bf16 *)s.dK; a.dV[g] = (__bf16 *)s.dV;
    AttnDims dm{B, H, s.Lq, s.Lk, 0, s.Lkp, s.q_bs, s.q_rs, s.q_hs, s.k_bs, s.k_rs, s.k_hs, s.o_bs, s.o_rs, s.o_hs, s.mask,
                scale, 1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), s.seed, seed_ptr, 0};
    dm.nkt1 = (s.Lk + AT_KB - 1) / AT_KB;
    a.dm[g] = dm;
  }
  return 0;
}

extern "C" __attribute__((visibility("default"))) int bq_attn_fwd_pair(const bq_attn_side *sides, int B, int H, float scale,
                                                                     float p_drop, const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(sides && B > 0 && H > 0, BQ_EINVAL, "attn_fwd_pair: bad arguments");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_fwd_pair: bad dropout probability");
  AttnPair a;
  int rc = fill_pair(sides, B, H, scale, p_drop, seed_ptr, false, a);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_fwd_narrow_pair_kernel, dim3(1, B * H, 2), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("attn_fwd_pair");
}

extern "C" __attribute__((visibility("default"))) int bq_attn_bwd_pair(const bq_attn_side *sides, int B, int H, float scale,
                                                                     float p_drop, const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(sides && B > 0 && H > 0, BQ_EINVAL, "attn_bwd_pair: bad arguments");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_bwd_pair: bad dropout probability");
  AttnPair a;
  int rc = fill_pair(sides, B, H, scale, p_drop, seed_ptr, true, a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_narrow_pair_kernel, dim3(1, B * H, 2), dim3(256), 0, st, a);
  rc = check_launch("attn_bwd_pair_dq");
  if (rc) return rc;
  const int nb0 = (sides[0].Lk + AT_QB - 1) / AT_QB, nb1 = (sides[1].Lk + AT_QB - 1) / AT_QB;
  hipLaunchKernelGGL(attn_bwd_dkv_pair_kernel, dim3(nb0 > nb1 ? nb0 : nb1, B * H, 2), dim3(256), 0, st, a);
  return check_launch("attn_bwd_pair_dkv");
}

// ---- attention probabilities on request ---------------------------------------------------------------------------------
// The reference returns the softmax matrix of a layer (before dropout) when asked (output_attentions, models/med.py:202,223;
// BLIP_VQA3D stores the last level's cross-attention maps in data_dict, blip_vqa_3d.py:262-281).  The fused kernels
// never form it; this kernel rebuilds it from Q, K and the forward's LSE: P[b][h][q][k] = exp2(c * q.k + mask[k] - LSE[q]),
// fp32 (B, H, Lq, Lk); with a drop threshold also the dropped map (x 1/(1-p) where the forward's hash kept the entry, 0
// elsewhere).  Thread = one key; the queries of a (batch, head) go through LDS 32 rows at a time.  A few MB for the
// text shapes it exists for.
__global__ __launch_bounds__(256) void attn_probs_kernel(const __bf16 *__restrict__ Q, const __bf16 *__restrict__ K,
                                                         const float *__restrict__ LSE, float *__restrict__ P, AttnDims dm) {
  __shared__ float s_q[32][64];
  __shared__ float s_l[32];
  const int bh = blockIdx.y, b = bh / dm.H, hd = bh % dm.H;
  const int key = blockIdx.x * 256 + threadIdx.x;
  const bool live = key < dm.Lk;
  const float c = dm.scale * 1.4426950408889634f;
  const unsigned seed = eff_seed(dm);
  float kr[64];
  {
    const __bf16 *kp = K + b * dm.k_bs + hd * dm.k_hs + (long)min(key, dm.Lk - 1) * dm.k_rs;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(kp + 8 * ch);
#pragma unroll
      for (int j = 0; j < 8; ++j) kr[8 * ch + j] = (float)v[j];
    }
  }
  const float mk = (dm.mask && live) ? dm.mask[(long)b * dm.Lkp + key] : 0.0f;
  for (int q0 = 0; q0 < dm.Lq; q0 += 32) {
    const int nq = min(32, dm.Lq - q0);
    __syncthreads();
    for (int e = threadIdx.x; e < nq * 64; e += 256)
      s_q[e >> 6][e & 63] = (float)Q[b * dm.q_bs + hd * dm.q_hs + (long)(q0 + (e >> 6)) * dm.q_rs + (e & 63)];
    if (threadIdx.x < nq) s_l[threadIdx.x] = LSE[(long)bh * dm.Lq + q0 + threadIdx.x];
    __syncthreads();
    if (!live) continue;
    for (int r = 0; r < nq; ++r) {
      float dot = 0.0f;
#pragma unroll
      for (int d = 0; d < 64; ++d) dot = __builtin_fmaf(s_q[r][d], kr[d], dot);
      const int q = q0 + r;
      float p = __builtin_amdgcn_exp2f(__builtin_fmaf(dot, c, mk) - s_l[r]);
      if (dm.causal && key > q) p = 0.0f;
      if (dm.drop_thresh) p = drop_keep(seed, bh, q, key, dm.drop_thresh) ? p * dm.inv_keep : 0.0f;
      P[((long)bh * dm.Lq + q) * dm.Lk + key] = p;
    }
  }
}

extern "C" __attribute__((visibility("default"))) int bq_attn_probs(
    const void *Q, const void *K, const float *LSE, const float *mask, float *P, int B, int H, int Lq, int Lk, int Lkp,
    long q_bs, long q_rs, long q_hs, long k_bs, long k_rs, long k_hs, float scale, float p_drop, unsigned seed,
    const unsigned *seed_ptr, int causal, void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0 && Q && K && LSE && P, BQ_EINVAL, "attn_probs: bad arguments");
  BQ_REQUIRE(!mask || (Lkp >= Lk), BQ_EINVAL, "attn_probs: mask row shorter than Lk");
  BQ_REQUIRE((k_rs % 8) == 0 && (k_hs % 8) == 0, BQ_EINVAL, "attn_probs: K rows must be 16-byte aligned");
  BQ_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, BQ_EINVAL, "attn_probs: bad dropout probability");
  AttnDims dm{B, H, Lq, Lk, 0, Lkp, q_bs, q_rs, q_hs, k_bs, k_rs, k_hs, 0, 0, 0, mask, scale,
              1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr, causal ? 1 : 0};
  hipLaunchKernelGGL(attn_probs_kernel, dim3((Lk + 255) / 256, B * H), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16 *)Q, (const __bf16 *)K, LSE, P, dm);
  return check_launch("attn_probs");
}
