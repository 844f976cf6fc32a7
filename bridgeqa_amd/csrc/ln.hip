// Fused  y = LayerNorm(dropout(x) + residual) * gamma + beta  for gfx950 -- the tail of BertSelfOutput /
// BertOutput (reference models/med.py:236-239, 313-317: dense -> dropout -> LayerNorm(h + input)), which the
// reference runs as dropout (2 kernels) + add + layer_norm (+ dtype casts) per site, 144 sites per step.
// One wave per row (hidden size = 64 * 4 * NCH, 768 -> NCH = 3), values stay in registers, statistics by
// DPP wave sums; the dropout mask is a stateless hash (regenerated in the backward, nothing stored).
// Also here: the padded transpose the attention kernels want (one launch instead of zeros + strided copy).
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
};

__device__ __forceinline__ unsigned ln_seed(const LnArgs &a) {
  return a.seed_ptr ? a.seed_ptr[0] * 2654435761u + a.seed : a.seed;
}

// z = dropout(x) + residual for the 4*NCH columns a lane owns: columns ch*256 + lane*4 + j
template <int NCH>
__device__ __forceinline__ void load_z(const __bf16 *x, const __bf16 *res, long rowoff, int row, int lane,
                                       const LnArgs &a, unsigned seed, float *z) {
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 256 + lane * 4;
    const bf16x4 xv = *reinterpret_cast<const bf16x4 *>(x + rowoff + c0);
    const bf16x4 rv = *reinterpret_cast<const bf16x4 *>(res + rowoff + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = (float)xv[j];
      if (a.thresh) v = ln_keep(seed, row, c0 + j, a.thresh) ? v * a.inv_keep : 0.0f;
      z[ch * 4 + j] = v + (float)rv[j];
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void drop_add_ln_fwd_kernel(const __bf16 *__restrict__ x,
                                                              const __bf16 *__restrict__ res,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, __bf16 *__restrict__ y,
                                                              float *__restrict__ mean_out,
                                                              float *__restrict__ rstd_out, LnArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wid;
  if (row >= a.M) return;
  const unsigned seed = ln_seed(a);
  const long rowoff = (long)row * a.H;
  float z[4 * NCH];
  load_z<NCH>(x, res, rowoff, row, lane, a, seed, z);
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
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// backward: rows are strided over the grid so every wave folds its rows' dgamma / dbeta in registers; one LDS
// reduction over the 4 waves and one atomicAdd per column per workgroup at the end
template <int NCH>
__global__ __launch_bounds__(256) void drop_add_ln_bwd_kernel(const __bf16 *__restrict__ x,
                                                              const __bf16 *__restrict__ res,
                                                              const float *__restrict__ gamma,
                                                              const __bf16 *__restrict__ dy,
                                                              const float *__restrict__ mean_in,
                                                              const float *__restrict__ rstd_in,
                                                              __bf16 *__restrict__ dx, __bf16 *__restrict__ dres,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                              LnArgs a) {
  __shared__ float s_g[4][256 * NCH], s_b[4][256 * NCH];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
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
  for (int row = blockIdx.x * 4 + wid; row < a.M; row += gridDim.x * 4) {
    const long rowoff = (long)row * a.H;
    float z[4 * NCH], g[4 * NCH];
    load_z<NCH>(x, res, rowoff, row, lane, a, seed, z);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const bf16x4 d = *reinterpret_cast<const bf16x4 *>(dy + rowoff + ch * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = ch * 4 + j;
        const float dyv = (float)d[j];
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
        const float dz = rstd * (g[i] - s1 - z[i] * s2);
        orr[j] = (__bf16)dz;
        float dxv = dz;
        if (a.thresh) dxv = ln_keep(seed, row, c0 + j, a.thresh) ? dz * a.inv_keep : 0.0f;
        ox[j] = (__bf16)dxv;
      }
      *reinterpret_cast<bf16x4 *>(dx + rowoff + c0) = ox;
      *reinterpret_cast<bf16x4 *>(dres + rowoff + c0) = orr;
    }
  }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s_g[wid][ch * 256 + lane * 4 + j] = ag[ch * 4 + j];
      s_b[wid][ch * 256 + lane * 4 + j] = ab[ch * 4 + j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < 256 * NCH; c += 256) {
    atomicAdd(dgamma + c, (s_g[0][c] + s_g[1][c]) + (s_g[2][c] + s_g[3][c]));
    atomicAdd(dbeta + c, (s_b[0][c] + s_b[1][c]) + (s_b[2][c] + s_b[3][c]));
  }
}

// out[bh][d][l] = in[b][l][h][d] for l < L, 0 for L <= l < Lp   (bf16; in given by element strides)
__global__ __launch_bounds__(256) void transpose_pad_kernel(const __bf16 *__restrict__ in, __bf16 *__restrict__ out,
                                                            int H, int L, int Lp, long bs, long rs, long hs) {
  __shared__ __bf16 tile[64][66];
  const int bh = blockIdx.y, b = bh / H, hd = bh % H;
  const int l0 = blockIdx.x * 64;
  const __bf16 *src = in + b * bs + hd * hs;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int l = i >> 6, d = i & 63;
    tile[l][d] = (l0 + l < L) ? src[(long)(l0 + l) * rs + d] : (__bf16)0.0f;
  }
  __syncthreads();
  __bf16 *dst = out + (long)bh * 64 * Lp + l0;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int d = i >> 6, l = i & 63;
    dst[(long)d * Lp + l] = tile[l][d];
  }
}

// Column sums of a bf16 (M, N) matrix into f32 (N): the bias gradient of every linear layer (db = sum_rows dY).
// Thread = 4 adjacent columns (8-B loads), 4 row phases per workgroup folded through LDS; row chunks over
// gridDim.y with atomics (out zero-initialised) only when M is large.
__global__ __launch_bounds__(256) void colsum_kernel(const __bf16 *__restrict__ g, float *__restrict__ out, int M,
                                                     int N, int rows_per_chunk) {
  __shared__ float s[4][256];
  const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + cg) * 4;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c0 < N) {
    for (int r = r0 + ph; r < r1; r += 4) {
      const bf16x4 v = *reinterpret_cast<const bf16x4 *>(g + (long)r * N + c0);
      a0 += (float)v[0]; a1 += (float)v[1]; a2 += (float)v[2]; a3 += (float)v[3];
    }
  }
  s[ph][cg * 4 + 0] = a0; s[ph][cg * 4 + 1] = a1; s[ph][cg * 4 + 2] = a2; s[ph][cg * 4 + 3] = a3;
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    const float v = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
    if (gridDim.y == 1) out[c] = v; else atomicAdd(out + c, v);
  }
}

}  // namespace bq

using namespace bq;

// out[n] = sum_m g[m][n]; g bf16 (M, N) row-major with N % 4 == 0, out f32 (N).  Returns in *needs_zero (host int,
// may be NULL) whether `out` had to be zero-initialised by the caller (large M: chunked with atomics) -- call
// bq_colsum_chunks first to know.
extern "C" __attribute__((visibility("default"))) int bq_colsum_chunks(int M) { return M <= 4096 ? 1 : (M + 1023) / 1024; }

extern "C" __attribute__((visibility("default"))) int bq_colsum_bf16(const void *g, float *out, int M, int N,
                                                                     void *stream) {
  BQ_REQUIRE(M >= 0 && N > 0 && N % 4 == 0, BQ_EINVAL, "colsum: bad extents");
  BQ_REQUIRE(g && out, BQ_EINVAL, "colsum: null pointer");
  const int chunks = bq_colsum_chunks(M);
  const int rpc = chunks == 1 ? (M > 0 ? M : 1) : 1024;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16 *)g, out, M, N, rpc);
  return check_launch("colsum");
}

// y = LayerNorm(dropout(x) + residual): x, residual, y bf16 (M, H) row-major, gamma/beta f32 (H), mean/rstd f32 (M)
// saved for the backward.  H must be 256, 512, 768 or 1024.
extern "C" __attribute__((visibility("default"))) int bq_drop_add_ln_fwd(
    const void *x, const void *residual, const float *gamma, const float *beta, void *y, float *mean, float *rstd,
    int M, int H, float eps, float p_drop, unsigned seed, const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(M >= 0 && H > 0 && H % 256 == 0 && H <= 1024, BQ_ELIMIT, "drop_add_ln: H=%d unsupported", H);
  if (M == 0) return BQ_OK;
  BQ_REQUIRE(x && residual && gamma && beta && y && mean && rstd, BQ_EINVAL, "drop_add_ln: null pointer");
  LnArgs a{M, H, eps, 1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr};
  const dim3 grid((M + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
#define BQ_LN_FWD(N)                                                                                           \
  hipLaunchKernelGGL(drop_add_ln_fwd_kernel<N>, grid, dim3(256), 0, st, (const __bf16 *)x, (const __bf16 *)residual, \
                     gamma, beta, (__bf16 *)y, mean, rstd, a)
  switch (H / 256) { case 1: BQ_LN_FWD(1); break; case 2: BQ_LN_FWD(2); break; case 3: BQ_LN_FWD(3); break; default: BQ_LN_FWD(4); }
#undef BQ_LN_FWD
  return check_launch("drop_add_ln_fwd");
}

// dgamma / dbeta (f32, H) MUST be zero-initialised; dx, dresidual bf16 (M, H).
extern "C" __attribute__((visibility("default"))) int bq_drop_add_ln_bwd(
    const void *x, const void *residual, const float *gamma, const void *dy, const float *mean, const float *rstd,
    void *dx, void *dresidual, float *dgamma, float *dbeta, int M, int H, float eps, float p_drop, unsigned seed,
    const unsigned *seed_ptr, void *stream) {
  BQ_REQUIRE(M >= 0 && H > 0 && H % 256 == 0 && H <= 1024, BQ_ELIMIT, "drop_add_ln: H=%d unsupported", H);
  if (M == 0) return BQ_OK;
  BQ_REQUIRE(x && residual && gamma && dy && mean && rstd && dx && dresidual && dgamma && dbeta, BQ_EINVAL,
             "drop_add_ln_bwd: null pointer");
  LnArgs a{M, H, eps, 1.0f / (1.0f - p_drop), (unsigned)((double)p_drop * 4294967296.0), seed, seed_ptr};
  int blocks = (M + 3) / 4;
  if (blocks > 512) blocks = 512;
  hipStream_t st = (hipStream_t)stream;
#define BQ_LN_BWD(N)                                                                                            \
  hipLaunchKernelGGL(drop_add_ln_bwd_kernel<N>, dim3(blocks), dim3(256), 0, st, (const __bf16 *)x,               \
                     (const __bf16 *)residual, gamma, (const __bf16 *)dy, mean, rstd, (__bf16 *)dx,              \
                     (__bf16 *)dresidual, dgamma, dbeta, a)
  switch (H / 256) { case 1: BQ_LN_BWD(1); break; case 2: BQ_LN_BWD(2); break; case 3: BQ_LN_BWD(3); break; default: BQ_LN_BWD(4); }
#undef BQ_LN_BWD
  return check_launch("drop_add_ln_bwd");
}

// in: bf16 (B, L, H, 64) by element strides -> out: bf16 [B*H][64][Lp], zero padded (Lp % 64 == 0)
extern "C" __attribute__((visibility("default"))) int bq_transpose_pad(const void *in, void *out, int B, int H, int L,
                                                                       int Lp, long bs, long rs, long hs,
                                                                       void *stream) {
  BQ_REQUIRE(B > 0 && H > 0 && L > 0 && Lp >= L && Lp % 64 == 0, BQ_EINVAL, "transpose_pad: bad extents");
  BQ_REQUIRE(in && out, BQ_EINVAL, "transpose_pad: null pointer");
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(Lp / 64, B * H), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16 *)in, (__bf16 *)out, H, L, Lp, bs, rs, hs);
  return check_launch("transpose_pad");
}
