// Fused  y = LayerNorm(dropout(x) + residual) * gamma + beta  for gfx950 -- the tail of BertSelfOutput /
// BertOutput (reference models/med.py:236-239, 313-317: dense -> dropout -> LayerNorm(h + input)), which the
// reference runs as dropout (2 kernels) + add + layer_norm (+ dtype casts) per site, 144 sites per step.
// One wave per row (hidden size = 64 * 4 * NCH, 768 -> NCH = 3), values stay in registers, statistics by
// DPP wave sums; the dropout mask is a stateless hash (regenerated in the backward, nothing stored).
#include <cstdint>
#include <stdlib.h>
#include "bq_common.h"

namespace bq {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_f32(float v) {
  asm volatile(BQ_DPP_WAVE("v_add_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ bool ln_keep(unsigned seed, int row, int col, unsigned thresh) {
  unsigned x = seed ^ ((unsigned)row * 0x85EBCA77u) ^ ((unsigned)col * 0xC2B2AE3Du);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x >= thresh;
}

struct LnArgs {
  int M, H;
  float eps, inv_keep;
  unsigned thresh, seed;
  const unsigned *seed_ptr;
  // stochastic depth (timm DropPath, reference models/vit.py:107-108): the whole x row of sample row / rows_per_sample
  // is dropped with probability path_thresh / 2^32 and scaled by path_inv_keep otherwise
  unsigned path_thresh;
  float path_inv_keep;
  int rows_per_sample;
  // row groups (the two text streams of the twin encoder stacked in one tensor, reference med.py:549-614): group g =
  // rows [g*M/groups, (g+1)*M/groups) with its OWN LayerNorm parameters (gamma2 / beta2 for g = 1) and its own dgamma /
  // dbeta accumulator (dgb + g*2H); blockIdx.y = group
  int groups;
  const float *gamma2, *beta2;
  // backward only (round 6): `x` IS the stored sum path(x) + residual (the forward's sum_out; no dropout): the row is read
  // once instead of twice, the gradients are dx = dz * path scale and dresidual = dz as before
  int x_is_sum;
};

__device__ __forceinline__ float ln_path_scale(const LnArgs &a, unsigned seed, int row) {
  if (!a.path_thresh) return 1.0f;
  return ln_keep(seed ^ 0x5bd1e995u, row / a.rows_per_sample, 0x3039, a.path_thresh) ? a.path_inv_keep : 0.0f;
}

__device__ __forceinline__ unsigned ln_seed(const LnArgs &a) {
  return a.seed_ptr ? a.seed_ptr[0] * 2654435761u + a.seed : a.seed;
}

// z = dropout(x) + residual for the 4*NCH columns a lane owns: columns ch*256 + lane*4 + j
template <int NCH>
__device__ __forceinline__ void load_z(const __bf16 *x, const __bf16 *res, long rowoff, int row, int lane,
                                       const LnArgs &a, unsigned seed, float *z) {
  const float ps = ln_path_scale(a, seed, row);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 256 + lane * 4;
    const bf16x4 xv = *reinterpret_cast<const bf16x4 *>(x + rowoff + c0);
    bf16x4 rv = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
    if (res) rv = *reinterpret_cast<const bf16x4 *>(res + rowoff + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = (float)xv[j];
      if (a.thresh) v = ln_keep(seed, row, c0 + j, a.thresh) ? v * a.inv_keep : 0.0f;
      z[ch * 4 + j] = v * ps + (float)rv[j];
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void drop_add_ln_fwd_kernel(const __bf16 *__restrict__ x,
                                                              const __bf16 *__restrict__ res,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, __bf16 *__restrict__ y,
                                                              __bf16 *__restrict__ sum_out,
                                                              float *__restrict__ mean_out,
                                                              float *__restrict__ rstd_out,
                                                              float *__restrict__ zero_out, LnArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int grp = blockIdx.y, Mg = a.M / a.groups;
  if (grp) { gamma = a.gamma2; beta = a.beta2; }
  if (zero_out && blockIdx.x == 0)  // the backward's dgamma / dbeta accumulators, cleared here for free
    for (int c = threadIdx.x; c < 2 * a.H; c += 256) zero_out[grp * 2 * a.H + c] = 0.0f;
  const unsigned seed = ln_seed(a);
  // (as in the backward: the next row's loads are in flight while this row is normalised)
  const bf16x4 zero4 = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
  bf16x4 cx[NCH], cr[NCH], nx[NCH], nr[NCH];
  auto fetch = [&](int row, bf16x4 (&fx)[NCH], bf16x4 (&fr)[NCH]) {
    const long rowoff = (long)row * a.H;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 256 + lane * 4;
      fx[ch] = *reinterpret_cast<const bf16x4 *>(x + rowoff + c0);
      fr[ch] = res ? *reinterpret_cast<const bf16x4 *>(res + rowoff + c0) : zero4;
    }
  };
  const int row_end = (grp + 1) * Mg, stride = gridDim.x * 4;
  int row0 = grp * Mg + blockIdx.x * 4 + wid;
  if (row0 < row_end) fetch(row0, cx, cr);
  for (int row = row0; row < row_end; row += stride) {
  const bool more = row + stride < row_end;
  if (more) fetch(row + stride, nx, nr);
  const long rowoff = (long)row * a.H;
  float z[4 * NCH];
  {
    const float ps = ln_path_scale(a, seed, row);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 256 + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = (float)cx[ch][j];
        if (a.thresh) v = ln_keep(seed, row, c0 + j, a.thresh) ? v * a.inv_keep : 0.0f;
        z[ch * 4 + j] = v * ps + (float)cr[ch][j];
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4 * NCH; ++i) s += z[i];
  const float mean = wave_sum_f32(s) / (float)a.H;
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < 4 * NCH; ++i) { const float d = z[i] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum_f32(v) / (float)a.H + a.eps);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4 *>(gamma + c0), bt = *reinterpret_cast<const float4 *>(beta + c0);
    bf16x4 o;
    o[0] = (__bf16)((z[ch * 4 + 0] - mean) * rstd * g.x + bt.x);
    o[1] = (__bf16)((z[ch * 4 + 1] - mean) * rstd * g.y + bt.y);
    o[2] = (__bf16)((z[ch * 4 + 2] - mean) * rstd * g.z + bt.z);
    o[3] = (__bf16)((z[ch * 4 + 3] - mean) * rstd * g.w + bt.w);
    *reinterpret_cast<bf16x4 *>(y + rowoff + c0) = o;
    if (sum_out) {  // the residual stream itself (pre-LN blocks carry it on)
      bf16x4 zs;
      zs[0] = (__bf16)z[ch * 4 + 0]; zs[1] = (__bf16)z[ch * 4 + 1]; zs[2] = (__bf16)z[ch * 4 + 2]; zs[3] = (__bf16)z[ch * 4 + 3];
      *reinterpret_cast<bf16x4 *>(sum_out + rowoff + c0) = zs;
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  if (more) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { cx[ch] = nx[ch]; cr[ch] = nr[ch]; }
  }
  }
}

// backward: rows are strided over the grid so every wave folds its rows' dgamma / dbeta in registers; one LDS
// reduction over the 4 waves and one atomicAdd per column per workgroup at the end into dgb (2, H), which the
// FORWARD launch of the same site zeroed (zero_out).  (A last-workgroup fold behind a release fence was measured
// 8x slower here: the fence has to write back the dx / dresidual lines this kernel just dirtied in L2.)
template <int NCH>
__global__ __launch_bounds__(256) void drop_add_ln_bwd_kernel(const __bf16 *__restrict__ x,
                                                              const __bf16 *__restrict__ res,
                                                              const float *__restrict__ gamma,
                                                              const __bf16 *__restrict__ dy,
                                                              const __bf16 *__restrict__ dsum,
                                                              const float *__restrict__ mean_in,
                                                              const float *__restrict__ rstd_in,
                                                              __bf16 *__restrict__ dx, __bf16 *__restrict__ dres,
                                                              float *__restrict__ dgb, LnArgs a) {
  __shared__ float s_g[4][256 * NCH], s_b[4][256 * NCH];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int grp = blockIdx.y, Mg = a.M / a.groups;
  if (grp) { gamma = a.gamma2; dgb += 2 * a.H; }
  const unsigned seed = ln_seed(a);
  float ag[4 * NCH], ab[4 * NCH], gm[4 * NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const float4 g = *reinterpret_cast<const float4 *>(gamma + ch * 256 + lane * 4);
    gm[ch * 4 + 0] = g.x; gm[ch * 4 + 1] = g.y; gm[ch * 4 + 2] = g.z; gm[ch * 4 + 3] = g.w;
  }
#pragma unroll
  for (int i = 0; i < 4 * NCH; ++i) { ag[i] = 0.0f; ab[i] = 0.0f; }
  const float invH = 1.0f / (float)a.H;
  // the loads of row r + stride are issued BEFORE row r is reduced (a wave otherwise has one row = 6 KB in flight and
  // waits a full memory round trip per row: 2.5 TB/s at the ViT shape)
  struct Raw {
    bf16x4 x[NCH], r[NCH], d[NCH], s[NCH];
    float mean, rstd;
  };
  const bf16x4 zero4 = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
  auto fetch = [&](int row, Raw &w) {
    const long rowoff = (long)row * a.H;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 256 + lane * 4;
      w.x[ch] = *reinterpret_cast<const bf16x4 *>(x + rowoff + c0);
      w.r[ch] = res ? *reinterpret_cast<const bf16x4 *>(res + rowoff + c0) : zero4;
      w.d[ch] = *reinterpret_cast<const bf16x4 *>(dy + rowoff + c0);
      w.s[ch] = dsum ? *reinterpret_cast<const bf16x4 *>(dsum + rowoff + c0) : zero4;
    }
    w.mean = mean_in[row];
    w.rstd = rstd_in[row];
  };
  const int row_end = (grp + 1) * Mg, stride = gridDim.x * 4;
  int row = grp * Mg + blockIdx.x * 4 + wid;
  Raw cur, nxt;
  if (row < row_end) fetch(row, cur);
  for (; row < row_end; row += stride) {
    const bool more = row + stride < row_end;  // wave-uniform
    if (more) fetch(row + stride, nxt);
    const long rowoff = (long)row * a.H;
    float z[4 * NCH], g[4 * NCH];
    const float ps = ln_path_scale(a, seed, row);
    const float mean = cur.mean, rstd = cur.rstd;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 256 + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ch * 4 + j;
        float v = (float)cur.x[ch][j];
        if (a.thresh) v = ln_keep(seed, row, c0 + j, a.thresh) ? v * a.inv_keep : 0.0f;
        z[i] = a.x_is_sum ? v : v * ps + (float)cur.r[ch][j];   // dropout(x) * path + residual, as load_z (or the stored sum)
        const float dyv = (float)cur.d[ch][j];
        z[i] = (z[i] - mean) * rstd;  // z_hat
        g[i] = dyv * gm[i];
        s1 += g[i];
        s2 += g[i] * z[i];
        ag[i] += dyv * z[i];
        ab[i] += dyv;
      }
    }
    s1 = wave_sum_f32(s1) * invH;
    s2 = wave_sum_f32(s2) * invH;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 256 + lane * 4;
      bf16x4 ox, orr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ch * 4 + j;
        const float dz = rstd * (g[i] - s1 - z[i] * s2) + (float)cur.s[ch][j];
        orr[j] = (__bf16)dz;
        float dxv = dz * ps;
        if (a.thresh) dxv = ln_keep(seed, row, c0 + j, a.thresh) ? dxv * a.inv_keep : 0.0f;
        ox[j] = (__bf16)dxv;
      }
      *reinterpret_cast<bf16x4 *>(dx + rowoff + c0) = ox;
      if (dres) *reinterpret_cast<bf16x4 *>(dres + rowoff + c0) = orr;
    }
    if (more) cur = nxt;
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s_g[wid][ch * 256 + lane * 4 + j] = ag[ch * 4 + j];
      s_b[wid][ch * 256 + lane * 4 + j] = ab[ch * 4 + j];
    }
  __syncthreads();
  constexpr int H = 256 * NCH;
  for (int c = threadIdx.x; c < H; c += 256) {
    atomicAdd(dgb + c, (s_g[0][c] + s_g[1][c]) + (s_g[2][c] + s_g[3][c]));
    atomicAdd(dgb + H + c, (s_b[0][c] + s_b[1][c]) + (s_b[2][c] + s_b[3][c]));
  }
}

// Column sums of a bf16 (M, N) matrix into f32 (N): the bias gradient of every linear layer (db = sum_rows dY).
// Workgroup = 256 columns x one chunk of rows; a lane owns VEC adjacent columns (16-B loads when VEC = 8) and the
// 256 / (256 / VEC) row phases are folded through LDS.  With more than one row chunk each workgroup stores its
// partial row to `partial` [chunks][N]; then either (counter != NULL, few workgroups) the LAST workgroup of a
// column block to finish (device-scope counter, reset to 0 on exit so the slot can be reused) adds the partials in
// chunk order -- one launch -- or (counter == NULL, large M: a release fence per workgroup means an L2 write-back
// on this multi-XCD part, measured 10x slower than the read itself) colsum_fold_kernel does it as a second launch.
// No zero-fill, no float atomics, bit-reproducible either way.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16 *__restrict__ g, float *__restrict__ out,
                                                     float *partial, unsigned *counter, int M, int N,
                                                     int rows_per_chunk) {
  constexpr int LANES = 256 / VEC, PH = 256 / LANES;
  typedef __bf16 vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ float s[PH][256];
  __shared__ int last;
  const int cg = threadIdx.x % LANES, ph = threadIdx.x / LANES;
  const int c0 = blockIdx.x * 256 + cg * VEC;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (c0 < N) {
    const __bf16 *p = g + (long)(r0 + ph) * N + c0;
    const long step = (long)PH * N;
    int r = r0 + ph;
    for (; r + 3 * PH < r1; r += 4 * PH, p += 4 * step) {
      const vec_t v0 = *reinterpret_cast<const vec_t *>(p), v1 = *reinterpret_cast<const vec_t *>(p + step),
                  v2 = *reinterpret_cast<const vec_t *>(p + 2 * step), v3 = *reinterpret_cast<const vec_t *>(p + 3 * step);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += ((float)v0[i] + (float)v1[i]) + ((float)v2[i] + (float)v3[i]);
    }
    for (; r < r1; r += PH, p += step) {
      const vec_t v = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += (float)v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[ph][cg * VEC + i] = acc[i];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
#pragma unroll
  for (int q = 0; q < PH; ++q) v += s[q][threadIdx.x];
  if (gridDim.y == 1) {
    if (c < N) out[c] = v;
    return;
  }
  if (c < N) partial[(long)blockIdx.y * N + c] = v;
  if (!counter) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter + blockIdx.x, 1u) == gridDim.y - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (c < N) {
    float t = 0.f;
    for (int y = 0; y < (int)gridDim.y; ++y) t += __builtin_nontemporal_load(partial + (long)y * N + c);
    out[c] = t;
  }
  if (threadIdx.x == 0) counter[blockIdx.x] = 0u;
}

// Small M (the text side: 160-1024 rows): one 1024-thread workgroup per 256 columns, 32 row phases, every load of a
// thread issued before the first add -- one launch, no partials, no fence (4 us instead of 9.6)
template <int VEC>
__global__ __launch_bounds__(1024) void colsum_tall_kernel(const __bf16 *__restrict__ g, float *__restrict__ out, int M,
                                                           int N) {
  constexpr int LANES = 256 / VEC, PH = 1024 / LANES;
  typedef __bf16 vec_t __attribute__((ext_vector_type(VEC)));
  __shared__ float s[PH][256 + 1];
  const int cg = threadIdx.x % LANES, ph = threadIdx.x / LANES;
  const int c0 = blockIdx.x * 256 + cg * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (c0 < N) {
    const __bf16 *p = g + (long)ph * N + c0;
    const long step = (long)PH * N;
    int r = ph;
    for (; r + 3 * PH < M; r += 4 * PH, p += 4 * step) {
      const vec_t v0 = *reinterpret_cast<const vec_t *>(p), v1 = *reinterpret_cast<const vec_t *>(p + step),
                  v2 = *reinterpret_cast<const vec_t *>(p + 2 * step), v3 = *reinterpret_cast<const vec_t *>(p + 3 * step);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += ((float)v0[i] + (float)v1[i]) + ((float)v2[i] + (float)v3[i]);
    }
    for (; r < M; r += PH, p += step) {
      const vec_t v = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += (float)v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[ph][cg * VEC + i] = acc[i];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
    for (int q = 0; q < PH; q += 4) {
      t0 += s[q][threadIdx.x]; t1 += s[q + 1][threadIdx.x]; t2 += s[q + 2][threadIdx.x]; t3 += s[q + 3][threadIdx.x];
    }
    if (c < N) out[c] = (t0 + t1) + (t2 + t3);
  }
}

// out[c] = sum_y partial[y][c] in a fixed order: 64 columns x 4 chunk phases per workgroup, 8 independent loads in
// flight per thread (L2 hits: latency-bound)
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                          int chunks, int N) {
  __shared__ float s[4][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float t[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] = 0.f;
  if (c < N) {
    int y = ph;
    for (; y + 28 < chunks; y += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] += partial[(long)(y + 4 * u) * N + c];
    }
    for (; y < chunks; y += 4) t[0] += partial[(long)y * N + c];
  }
  s[ph][cl] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  if (ph == 0 && c < N) out[c] = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
}

// exact (erf) GELU, bf16 in / out, fp32 arithmetic: y = 0.5 x (1 + erf(x / sqrt 2)) -- vit.py act_layer=nn.GELU and
// med_config "hidden_act": "gelu".  16-B loads / stores, 4 vectors in flight per lane.
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const __bf16 *__restrict__ x, __bf16 *__restrict__ y, long n8) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + i * 8);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = (float)v[j];
      o[j] = (__bf16)(0.5f * f * (1.0f + erff(f * 0.70710678118654752440f)));
    }
    *reinterpret_cast<bf16x8 *>(y + i * 8) = o;
  }
}

}  // namespace bq
using namespace bq;

// y = GELU(x) (exact), n bf16 elements, n % 8 == 0, 16-B aligned
extern "C" __attribute__((visibility("default"))) int bq_gelu_fwd_bf16(const void *x, void *y, long n, void *stream) {
  BQ_REQUIRE(n >= 0 && n % 8 == 0, BQ_EINVAL, "gelu: n must be a multiple of 8");
  if (n == 0) return BQ_OK;
  BQ_REQUIRE(x && y && (((uintptr_t)x | (uintptr_t)y) & 15) == 0, BQ_EINVAL, "gelu: null / unaligned pointer");
  const long n8 = n / 8;
  long blocks = (n8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                     (__bf16 *)y, n8);
  return check_launch("gelu_fwd");
}

// out[n] = sum_m g[m][n]; g bf16 (M, N) row-major with N % 4 == 0, out f32 (N), every element written.
// bq_colsum_chunks(M) = C row chunks: when C > 1 the caller passes `partial` (C * N floats, uninitialised), and for
// M <= BQ_COLSUM_ONE_LAUNCH_ROWS also `counter` ((N + 255) / 256 unsigned ints that are ZERO on entry; the kernel
// leaves them zero) -- counters of launches that may run concurrently must not alias.  Larger M: two launches.
#define BQ_COLSUM_ONE_LAUNCH_ROWS 2048
#define BQ_COLSUM_TALL_ROWS 1024
extern "C" __attribute__((visibility("default"))) int bq_colsum_chunks(int M) {
  if (M <= BQ_COLSUM_TALL_ROWS) return 1;
  const int rpc = M <= BQ_COLSUM_ONE_LAUNCH_ROWS ? 32 : 128;
  return M <= rpc ? 1 : (M + rpc - 1) / rpc;
}

extern "C" __attribute__((visibility("default"))) int bq_colsum_bf16(const void *g, float *out, int M, int N,
                                                                     float *partial, unsigned *counter, void *stream) {
  BQ_REQUIRE(M >= 0 && N > 0 && N % 4 == 0, BQ_EINVAL, "colsum: bad extents");
  BQ_REQUIRE((g || M == 0) && out, BQ_EINVAL, "colsum: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (M <= BQ_COLSUM_TALL_ROWS) {
    const dim3 grid((N + 255) / 256);
    if (N % 8 == 0)
      hipLaunchKernelGGL(colsum_tall_kernel<8>, grid, dim3(1024), 0, st, (const __bf16 *)g, out, M, N);
    else
      hipLaunchKernelGGL(colsum_tall_kernel<4>, grid, dim3(1024), 0, st, (const __bf16 *)g, out, M, N);
    return check_launch("colsum");
  }
  const int chunks = bq_colsum_chunks(M);
  const bool one_launch = M <= BQ_COLSUM_ONE_LAUNCH_ROWS;
  BQ_REQUIRE(chunks == 1 || (partial && (counter || !one_launch)), BQ_EINVAL, "colsum: workspace missing for %d chunks",
             chunks);
  const int rpc = chunks == 1 ? (M > 0 ? M : 1) : (one_launch ? 32 : 128);
  const dim3 grid((N + 255) / 256, chunks);
  unsigned *cnt = one_launch ? counter : nullptr;
  if (N % 8 == 0)
    hipLaunchKernelGGL(colsum_kernel<8>, grid, dim3(256), 0, st, (const __bf16 *)g, out, partial, cnt, M, N, rpc);
  else
    hipLaunchKernelGGL(colsum_kernel<4>, grid, dim3(256), 0, st, (const __bf16 *)g, out, partial, cnt, M, N, rpc);
  if (chunks > 1 && !one_launch)
    hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 63) / 64), dim3(256), 0, st, partial, out, chunks, N);
  return check_launch("colsum");
}

// y = LayerNorm(dropout(x) + residual): x, residual, y bf16 (M, H) row-major, gamma/beta f32 (H), mean/rstd f32 (M)
// saved for the backward.  H must be 256, 512, 768 or 1024.  residual may be NULL (plain LayerNorm(dropout(x)));
// sum_out (bf16 (M, H)) may be NULL, else it receives dropout(x) + residual -- the carried residual stream of a
// pre-LN block (reference models/vit.py:106-109).  zero_out (f32 (2, H), may be NULL) is set to 0: pass the buffer
// that bq_drop_add_ln_bwd of this site will accumulate dgamma / dbeta into.
static int ln_fwd_launch(const void *x, const void *residual, const float *gamma, const float *beta, const float *gamma2,
                         const float *beta2, int groups, void *y, void *sum_out, float *mean, float *rstd,
                         float *zero_out, int M, int H, float eps, float p_drop, float p_path, int rows_per_sample,
                         unsigned seed, const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(M >= 0 && H > 0 && H % 256 == 0 && H <= 1024, BQ_ELIMIT, "drop_add_ln: H=%d unsupported", H);
  BQ_REQUIRE(groups == 1 || (groups == 2 && gamma2 && beta2 && M % 2 == 0), BQ_EINVAL, "drop_add_ln: bad row groups");
  if (M == 0) {
    if (zero_out) (void)hipMemsetAsync(zero_out, 0, sizeof(float) * 2 * H * groups, (hipStream_t)stream);
    return check_launch("drop_add_ln_fwd");
  }
  BQ_REQUIRE(x && gamma && beta && y && mean && rstd, BQ_EINVAL, "drop_add_ln: null pointer");
  BQ_REQUIRE(p_path == 0.0f || rows_per_sample > 0, BQ_EINVAL, "drop_add_ln: rows_per_sample");
  LnArgs a{M, H, eps, 1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr,
           (unsigned)((double)p_path * 4294967296.0), 1.0f / (1.0f - p_path), rows_per_sample, groups, gamma2, beta2, 0};
#ifndef BQ_LN_FWD_CAP
#define BQ_LN_FWD_CAP 1024
#endif
  constexpr int fwd_cap = BQ_LN_FWD_CAP;  // rows are strided over the grid (tools/ln_sweep.sh)
  const int Mg = M / groups;
  const dim3 grid(((Mg + 3) / 4) < fwd_cap ? (Mg + 3) / 4 : fwd_cap, groups);
  hipStream_t st = (hipStream_t)stream;
#define BQ_LN_FWD(N)                                                                                           \
  hipLaunchKernelGGL(drop_add_ln_fwd_kernel<N>, grid, dim3(256), 0, st, (const __bf16 *)x, (const __bf16 *)residual, \
                     gamma, beta, (__bf16 *)y, (__bf16 *)sum_out, mean, rstd, zero_out, a)
  switch (H / 256) { case 1: BQ_LN_FWD(1); break; case 2: BQ_LN_FWD(2); break; case 3: BQ_LN_FWD(3); break; default: BQ_LN_FWD(4); }
#undef BQ_LN_FWD
  return check_launch("drop_add_ln_fwd");
}
extern "C" __attribute__((visibility("default"))) int bq_drop_add_ln_fwd(
    const void *x, const void *residual, const float *gamma, const float *beta, void *y, void *sum_out, float *mean,
    float *rstd, float *zero_out, int M, int H, float eps, float p_drop, float p_path, int rows_per_sample,
    unsigned seed, const unsigned *seed_ptr, void *stream) {
  return ln_fwd_launch(x, residual, gamma, beta, nullptr, nullptr, 1, y, sum_out, mean, rstd, zero_out, M, H, eps, p_drop,
                       p_path, rows_per_sample, seed, seed_ptr, stream);
}
// The same over TWO row groups with their own LayerNorm parameters: rows [0, M/2) use gamma / beta, rows [M/2, M) gamma2 /
// beta2 (the 2D and the 3D text stream of the twin encoder stacked in one tensor); zero_out: f32 (2, 2, H).
extern "C" __attribute__((visibility("default"))) int bq_twin_drop_add_ln_fwd(
    const void *x, const void *residual, const float *gamma, const float *beta, const float *gamma2, const float *beta2,
    void *y, float *mean, float *rstd, float *zero_out, int M, int H, float eps, float p_drop, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  return ln_fwd_launch(x, residual, gamma, beta, gamma2, beta2, 2, y, nullptr, mean, rstd, zero_out, M, H, eps, p_drop,
                       0.0f, 0, seed, seed_ptr, stream);
}

// dgb: f32 (2, H) = dgamma then dbeta, ACCUMULATED into (float atomics): it must hold zeros on entry -- the
// forward's zero_out does that.  dx, dresidual bf16 (M, H).  residual / dresidual NULL together for the plain
// form; dsum (bf16 (M, H), may be NULL) is the gradient that reached sum_out and is added to both.
static int ln_bwd_launch(const void *x, const void *residual, const float *gamma, const float *gamma2, int groups,
                         const void *dy, const void *dsum, const float *mean, const float *rstd, void *dx,
                         void *dresidual, float *dgb, int M, int H, float eps, float p_drop, float p_path,
                         int rows_per_sample, unsigned seed, const unsigned *seed_ptr, void *stream, int x_is_sum = 0) {
  BQ_REQUIRE(M >= 0 && H > 0 && H % 256 == 0 && H <= 1024, BQ_ELIMIT, "drop_add_ln: H=%d unsupported", H);
  BQ_REQUIRE(groups == 1 || (groups == 2 && gamma2 && M % 2 == 0), BQ_EINVAL, "drop_add_ln_bwd: bad row groups");
  if (M == 0) return BQ_OK;
  BQ_REQUIRE(x && gamma && dy && mean && rstd && dx && dgb && (x_is_sum ? !residual : (!residual == !dresidual)), BQ_EINVAL,
             "drop_add_ln_bwd: null pointer");
  BQ_REQUIRE(!x_is_sum || (p_drop == 0.0f && groups == 1), BQ_EINVAL, "drop_add_ln_bwd_sum: no dropout, one row group");
  BQ_REQUIRE(p_path == 0.0f || rows_per_sample > 0, BQ_EINVAL, "drop_add_ln_bwd: rows_per_sample");
  LnArgs a{M, H, eps, 1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr,
           (unsigned)((double)p_path * 4294967296.0), 1.0f / (1.0f - p_path), rows_per_sample, groups, gamma2, nullptr, x_is_sum};
  // (with the cross-row prefetch: 256 / 384 / 512 / 640 / 1024 / 2048 workgroups -> 33.1 / 32.2 / 35.2 / 40.6 / 41.6 / 60.6 us
  // at the ViT shape, tools/bench_ln.py: every workgroup ends with 2 H float atomics on the same 2 H addresses)
  constexpr int bwd_cap = 384;
  int blocks = (M / groups + 3) / 4;
  if (blocks > bwd_cap) blocks = bwd_cap;
  hipStream_t st = (hipStream_t)stream;
#define BQ_LN_BWD(N)                                                                                            \
  hipLaunchKernelGGL(drop_add_ln_bwd_kernel<N>, dim3(blocks, groups), dim3(256), 0, st, (const __bf16 *)x,       \
                     (const __bf16 *)residual, gamma, (const __bf16 *)dy, (const __bf16 *)dsum, mean, rstd,      \
                     (__bf16 *)dx, (__bf16 *)dresidual, dgb, a)
  switch (H / 256) { case 1: BQ_LN_BWD(1); break; case 2: BQ_LN_BWD(2); break; case 3: BQ_LN_BWD(3); break; default: BQ_LN_BWD(4); }
#undef BQ_LN_BWD
  return check_launch("drop_add_ln_bwd");
}
extern "C" __attribute__((visibility("default"))) int bq_drop_add_ln_bwd(
    const void *x, const void *residual, const float *gamma, const void *dy, const void *dsum, const float *mean,
    const float *rstd, void *dx, void *dresidual, float *dgb, int M, int H, float eps, float p_drop, float p_path,
    int rows_per_sample, unsigned seed, const unsigned *seed_ptr, void *stream) {
  return ln_bwd_launch(x, residual, gamma, nullptr, 1, dy, dsum, mean, rstd, dx, dresidual, dgb, M, H, eps, p_drop, p_path,
                       rows_per_sample, seed, seed_ptr, stream);
}
// ABI 6.  The backward of a site WITHOUT dropout whose forward wrote sum_out (the residual stream of a pre-LN block, reference
// models/vit.py:106-109), from that stored sum: `sum` bf16 (M, H) replaces x and residual (one read instead of two: the ViT
// site drops from 150 to 125 MB, to 100 MB without stochastic depth); dx = dz * path scale, dresidual = dz (NULL with
// p_path == 0: the two are the same tensor then and the caller hands dx to both).  The normalised value is re-formed from the
// bf16-rounded sum where bq_drop_add_ln_bwd re-forms it from x and residual in fp32 (one bf16 rounding of the row apart).
extern "C" __attribute__((visibility("default"))) int bq_drop_add_ln_bwd_sum(
    const void *sum, const float *gamma, const void *dy, const void *dsum, const float *mean, const float *rstd, void *dx,
    void *dresidual, float *dgb, int M, int H, float eps, float p_path, int rows_per_sample, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  return ln_bwd_launch(sum, nullptr, gamma, nullptr, 1, dy, dsum, mean, rstd, dx, dresidual, dgb, M, H, eps, 0.0f, p_path,
                       rows_per_sample, seed, seed_ptr, stream, 1);
}
// backward of bq_twin_drop_add_ln_fwd: dgb f32 (2, 2, H) = per row group dgamma then dbeta, accumulated
extern "C" __attribute__((visibility("default"))) int bq_twin_drop_add_ln_bwd(
    const void *x, const void *residual, const float *gamma, const float *gamma2, const void *dy, const float *mean,
    const float *rstd, void *dx, void *dresidual, float *dgb, int M, int H, float eps, float p_drop, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  return ln_bwd_launch(x, residual, gamma, gamma2, 2, dy, nullptr, mean, rstd, dx, dresidual, dgb, M, H, eps, p_drop, 0.0f,
                       0, seed, seed_ptr, stream);
}

