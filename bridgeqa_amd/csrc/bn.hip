// Training-mode BatchNorm + ReLU (+ max over the nsample axis) for the POINT-MAJOR detector path, gfx950.
//
// Replaces, per SharedMLP layer of a set-abstraction module (reference lib/pointnet2/pytorch_utils.py:104-157
// conv -> BatchNorm2d -> ReLU, and pointnet2_modules.py:259-262 F.max_pool2d(kernel=[1, nsample]) after the last
// layer), what runs today as MIOpen BatchNorm (3 kernels) + clamp + a generic max reduction forward and
// threshold_backward + 3 MIOpen kernels (+ the max's index scatter) backward: the activations are read
// 5 times forward and 6 times backward there; here 2 + 2.
//
// x: bf16 (R, C) rows = (b, m, s) point-major = the NHWC convolution output as it lies in memory, C % 8 == 0.
//   forward  pass 1  bn_stats_kernel      per-channel sum / sum of squares over row chunks -> partial [chunks][2C]
//            pass 1b bn_finalize_kernel   fold in chunk order; mean, biased var -> scale = g*rstd, shift = b - mean*scale;
//                                         running stats (unbiased var, momentum), num_batches_tracked += 1
//            pass 2  bn_apply_kernel      y = relu(x*scale + shift)  [or  out[b,m,:] = max_s of that]
//   backward pass 1  bn_bwd_reduce_kernel g = dy * [x*scale+shift > 0] (routed to the first arg-max row when pooled);
//                                         partial sums of g and g*xhat
//            pass 1b bn_fold_kernel       dbeta, dgamma
//            pass 2  bn_bwd_dx_kernel     dx = scale * (g - dbeta/R - xhat * dgamma/R)
// Statistics, scale/shift and all sums are fp32; the fixed chunk order makes every result bit-reproducible.
#include <cstdint>
#include "bq_common.h"

namespace bq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// rows per partial (measurement macros).  Round 5: 4096 / 1024 left the detector's large reductions with 512 workgroups -- eight
// waves per CU for a streaming kernel; c2 step, two runs each: 4096 / 1024 9.57 ms, 2048 / 512 9.16, 1024 / 512 9.19, 1024 / 256
// 9.07, 1024 / 128 9.06, 512 / 256 9.14-9.26, 512 / 128 9.14, 256 / 128 9.27 (more partials to fold)
#ifndef BQ_BN_CHUNK_BIG
#define BQ_BN_CHUNK_BIG 1024
#endif
#ifndef BQ_BN_CHUNK_SMALL
#define BQ_BN_CHUNK_SMALL 256
#endif
__host__ __device__ inline int bn_chunk_rows(long R) { return R >= (1L << 20) ? BQ_BN_CHUNK_BIG : BQ_BN_CHUNK_SMALL; }

// thread = 8 adjacent channels; TPR = C / 8 threads per row, 256 / TPR rows per sweep
__global__ __launch_bounds__(256) void bn_stats_kernel(const __bf16 *__restrict__ x, float *__restrict__ partial,
                                                       long R, int C) {
  __shared__ float s_sum[256 * 8], s_sq[256 * 8];
  const int tpr = C >> 3, rps = 256 / tpr;
  const int cg = threadIdx.x % tpr, rp = threadIdx.x / tpr;
  const int cr = bn_chunk_rows(R);
  const long r0 = (long)blockIdx.x * cr, r1 = min(R, r0 + cr);
  // sums are taken of (x - pivot) with pivot = row 0 of the same channel: E[x^2] - mean^2 in fp32 loses the variance
  // when |mean| >> std (e.g. mean 30, std 0.5 over 2 M rows); shifted by a value within a few std of the mean it does not
  float a[8], q[8], pv[8];
  {
    const bf16x8 p0 = *reinterpret_cast<const bf16x8 *>(x + cg * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = 0.f; q[i] = 0.f; pv[i] = (float)p0[i]; }
  }
  if (rp < rps) {
    for (long r = r0 + rp; r < r1; r += rps) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + r * C + cg * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float f = (float)v[i] - pv[i]; a[i] += f; q[i] += f * f; }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { s_sum[threadIdx.x * 8 + i] = a[i]; s_sq[threadIdx.x * 8 + i] = q[i]; }
  __syncthreads();
  // channel c = cg*8 + i lives at [(rp*tpr + cg)*8 + i]: fold the rps row phases in order
  for (int c = threadIdx.x; c < C; c += 256) {
    float ts = 0.f, tq = 0.f;
    for (int p = 0; p < rps; ++p) { ts += s_sum[(p * tpr) * 8 + c]; tq += s_sq[(p * tpr) * 8 + c]; }
    partial[(long)blockIdx.x * 2 * C + c] = ts;
    partial[(long)blockIdx.x * 2 * C + C + c] = tq;
  }
}

// out[j] = sum over chunks of partial[chunk][j], j < W, fixed order; 64 columns x 4 chunk phases per workgroup,
// 8 independent loads in flight per thread (the loads are L2 hits: latency, not bandwidth, is the cost)
__device__ __forceinline__ float fold_column(const float *partial, int chunks, int W, int j, int ph, float (*s)[64],
                                             int cl) {
  float t[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] = 0.f;
  int y = ph;
  for (; y + 28 < chunks; y += 32) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += partial[(long)(y + 4 * u) * W + j];
  }
  for (; y < chunks; y += 4) t[0] += partial[(long)y * W + j];
  s[ph][cl] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  const float v = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
  __syncthreads();
  return v;
}

struct BnParams {
  const float *gamma, *beta;
  float *running_mean, *running_var;  // may be null
  long long *num_batches_tracked;     // may be null
  float *scale, *shift, *mean, *rstd; // outputs, C each
  float eps, momentum;
};

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ partial,
                                                          const __bf16 *__restrict__ x, int chunks, int C, long R,
                                                          BnParams p) {
  __shared__ float s[4][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int cc = min(c, C - 1);
  const float sum = fold_column(partial, chunks, 2 * C, cc, ph, s, cl);
  const float sq = fold_column(partial, chunks, 2 * C, C + cc, ph, s, cl);
  if (ph != 0 || c >= C) return;
  const float invR = 1.0f / (float)R;
  const float dmean = sum * invR;  // mean of (x - pivot)
  const float mean = (float)x[c] + dmean;
  const float var = fmaxf(sq * invR - dmean * dmean, 0.0f);
  const float rstd = rsqrtf(var + p.eps);
  const float sc = p.gamma[c] * rstd;
  p.scale[c] = sc;
  p.shift[c] = p.beta[c] - mean * sc;
  p.mean[c] = mean;
  p.rstd[c] = rstd;
  if (p.running_mean) {
    const float unbiased = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
    p.running_mean[c] = (1.0f - p.momentum) * p.running_mean[c] + p.momentum * mean;
    p.running_var[c] = (1.0f - p.momentum) * p.running_var[c] + p.momentum * unbiased;
  }
  if (c == 0 && p.num_batches_tracked) p.num_batches_tracked[0] += 1;
}

// W-column fold of generic partials (backward: W = 2C -> dbeta | dgamma): 16 columns x 16 chunk phases per workgroup, 8 loads in
// flight per thread, the phases summed in a fixed order (round 5: 64 columns x 4 phases left a thread 512 dependent-batch loads
// of the 2048 partials the finer reduction chunks produce -- 10 us per layer)
constexpr int FOLD_COLS = 16;
__global__ __launch_bounds__(256) void bn_fold_kernel(const float *__restrict__ partial, float *__restrict__ out,
                                                      int chunks, int W) {
  __shared__ float s[16][FOLD_COLS];
  const int cl = threadIdx.x & (FOLD_COLS - 1), ph = threadIdx.x / FOLD_COLS;
  const int j = blockIdx.x * FOLD_COLS + cl, jj = min(j, W - 1);
  float t[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) t[u] = 0.f;
  int y = ph;
  for (; y + 16 * 7 < chunks; y += 16 * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += partial[(long)(y + 16 * u) * W + jj];
  }
  for (; y < chunks; y += 16) t[0] += partial[(long)y * W + jj];
  s[ph][cl] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  __syncthreads();
  if (ph == 0 && j < W) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += s[q][cl];
    out[j] = v;
  }
}

// y = relu?(x*scale + shift); POOL: out[g][c] = max over the S rows of group g (relu applied: max >= 0 when RELU)
// arg (POOL, may be null): u8 (R / S, C), the row 0 .. S-1 of the group's FIRST maximum -- what the backward searches for
// otherwise (bn_bwd_reduce_kernel / bn_bwd_dx_kernel); csrc/detbwd.hip reads the table instead
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bn_apply_kernel(const __bf16 *__restrict__ x, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, __bf16 *__restrict__ y, long R,
                                                       int C, int S, unsigned char *__restrict__ arg) {
  const int tpr = C >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(t % tpr);
  const long row = t / tpr;  // POOL: group index
  const long nrows = POOL ? R / S : R;
  if (row >= nrows) return;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i += 4) {
    const float4 a = *reinterpret_cast<const float4 *>(scale + cg * 8 + i), b = *reinterpret_cast<const float4 *>(shift + cg * 8 + i);
    sc[i] = a.x; sc[i + 1] = a.y; sc[i + 2] = a.z; sc[i + 3] = a.w;
    sh[i] = b.x; sh[i + 1] = b.y; sh[i + 2] = b.z; sh[i + 3] = b.w;
  }
  if (!POOL) {
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + row * C + cg * 8);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f = (float)v[i] * sc[i] + sh[i];
      if (RELU) f = fmaxf(f, 0.0f);
      o[i] = (__bf16)f;
    }
    *reinterpret_cast<bf16x8 *>(y + row * C + cg * 8) = o;
  } else {
    float m[8];
    int am[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = -INFINITY; am[i] = 0; }
    const __bf16 *px = x + row * S * C + cg * 8;
    for (int s = 0; s < S; ++s) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + (long)s * C);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = (float)v[i] * sc[i] + sh[i];
        if (z > m[i]) { m[i] = z; am[i] = s; }
      }
    }
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)(RELU ? fmaxf(m[i], 0.0f) : m[i]);
    *reinterpret_cast<bf16x8 *>(y + row * C + cg * 8) = o;
    if (arg) {
      uint2 pk;
      pk.x = (unsigned)am[0] | ((unsigned)am[1] << 8) | ((unsigned)am[2] << 16) | ((unsigned)am[3] << 24);
      pk.y = (unsigned)am[4] | ((unsigned)am[5] << 8) | ((unsigned)am[6] << 16) | ((unsigned)am[7] << 24);
      *reinterpret_cast<uint2 *>(arg + row * C + cg * 8) = pk;
    }
  }
}

// Backward pass 1: partial sums of g and g*xhat per channel.  POOL: dy is (R/S, C); the gradient of group g goes to
// the FIRST row attaining the maximum (and only if that maximum is > 0 under RELU).
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x,
                                                            const float *__restrict__ scale,
                                                            const float *__restrict__ shift,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, float *__restrict__ partial,
                                                            long R, int C, int S) {
  __shared__ float s_b[256 * 8], s_g[256 * 8];
  const int tpr = C >> 3, rps = 256 / tpr;
  const int cg = threadIdx.x % tpr, rp = threadIdx.x / tpr;
  float sc[8], sh[8], mu[8], rs[8], ab[8], ag[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[cg * 8 + i]; sh[i] = shift[cg * 8 + i]; mu[i] = mean[cg * 8 + i]; rs[i] = rstd[cg * 8 + i];
    ab[i] = 0.f; ag[i] = 0.f;
  }
  if (rp < rps) {
    if (!POOL) {
      const int cr = bn_chunk_rows(R);
      const long r0 = (long)blockIdx.x * cr, r1 = min(R, r0 + cr);
      for (long r = r0 + rp; r < r1; r += rps) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + r * C + cg * 8);
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dy + r * C + cg * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xf = (float)v[i];
          float g = (float)d[i];
          if (RELU && !(xf * sc[i] + sh[i] > 0.0f)) g = 0.0f;
          ab[i] += g;
          ag[i] += g * ((xf - mu[i]) * rs[i]);
        }
      }
    } else {
      const long G = R / S, gpc = bn_chunk_rows(R) / S > 0 ? bn_chunk_rows(R) / S : 1;
      const long g0 = (long)blockIdx.x * gpc, g1 = min(G, g0 + gpc);
      for (long gi = g0 + rp; gi < g1; gi += rps) {
        const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dy + gi * C + cg * 8);
        float best[8], bx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; bx[i] = 0.f; }
        const __bf16 *px = x + gi * S * C + cg * 8;
        for (int s = 0; s < S; ++s) {
          const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + (long)s * C);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float xf = (float)v[i], z = xf * sc[i] + sh[i];
            if (z > best[i]) { best[i] = z; bx[i] = xf; }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float g = (float)d[i];
          if (RELU && !(best[i] > 0.0f)) g = 0.0f;
          ab[i] += g;
          ag[i] += g * ((bx[i] - mu[i]) * rs[i]);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { s_b[threadIdx.x * 8 + i] = ab[i]; s_g[threadIdx.x * 8 + i] = ag[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float tb = 0.f, tg = 0.f;
    for (int p = 0; p < rps; ++p) { tb += s_b[(p * tpr) * 8 + c]; tg += s_g[(p * tpr) * 8 + c]; }
    partial[(long)blockIdx.x * 2 * C + c] = tb;
    partial[(long)blockIdx.x * 2 * C + C + c] = tg;
  }
}

// Backward pass 2: dx = scale * (g - dbeta/R - xhat * dgamma/R), every row of x gets a value
template <bool RELU, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                        const float *__restrict__ dgb, __bf16 *__restrict__ dx, long R,
                                                        int C, int S) {
  const int tpr = C >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int cg = (int)(t % tpr);
  const long row = t / tpr;  // POOL: group index
  const long nrows = POOL ? R / S : R;
  if (row >= nrows) return;
  const float invR = 1.0f / (float)R;
  float sc[8], sh[8], mu[8], rs[8], mb[8], mg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[cg * 8 + i]; sh[i] = shift[cg * 8 + i]; mu[i] = mean[cg * 8 + i]; rs[i] = rstd[cg * 8 + i];
    mb[i] = dgb[cg * 8 + i] * invR; mg[i] = dgb[C + cg * 8 + i] * invR;
  }
  if (!POOL) {
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + row * C + cg * 8);
    const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dy + row * C + cg * 8);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xf = (float)v[i];
      float g = (float)d[i];
      if (RELU && !(xf * sc[i] + sh[i] > 0.0f)) g = 0.0f;
      o[i] = (__bf16)(sc[i] * (g - mb[i] - (xf - mu[i]) * rs[i] * mg[i]));
    }
    *reinterpret_cast<bf16x8 *>(dx + row * C + cg * 8) = o;
  } else {
    const bf16x8 d = *reinterpret_cast<const bf16x8 *>(dy + row * C + cg * 8);
    const __bf16 *px = x + row * S * C + cg * 8;
    __bf16 *pd = dx + row * S * C + cg * 8;
    float best[8];
    int arg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = -INFINITY; arg[i] = 0; }
    for (int s = 0; s < S; ++s) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + (long)s * C);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = (float)v[i] * sc[i] + sh[i];
        if (z > best[i]) { best[i] = z; arg[i] = s; }
      }
    }
    float gsel[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gsel[i] = (RELU && !(best[i] > 0.0f)) ? 0.0f : (float)d[i];
    for (int s = 0; s < S; ++s) {
      const bf16x8 v = *reinterpret_cast<const bf16x8 *>(px + (long)s * C);
      bf16x8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float g = (arg[i] == s) ? gsel[i] : 0.0f;
        o[i] = (__bf16)(sc[i] * (g - mb[i] - ((float)v[i] - mu[i]) * rs[i] * mg[i]));
      }
      *reinterpret_cast<bf16x8 *>(pd + (long)s * C) = o;
    }
  }
}

}  // namespace bq
using namespace bq;

static bool bn_extents_ok(long R, int C, int S, int pool) {
  return R >= 0 && (C == 8 || C == 16 || C == 32 || C == 64 || C == 128 || C == 256 || C == 512 || C == 1024 || C == 2048) &&
         (!pool || (S > 0 && R % S == 0 && (1024 % S == 0 || S > 4096)));
}

extern "C" __attribute__((visibility("default"))) int bq_bn_chunks(long R, int S, int pool) {
  if (R <= 0) return 0;
  if (pool && S > 0) {
    const long G = R / S, gpc = bn_chunk_rows(R) / S > 0 ? bn_chunk_rows(R) / S : 1;
    return (int)((G + gpc - 1) / gpc);
  }
  return (int)((R + bn_chunk_rows(R) - 1) / bn_chunk_rows(R));
}

// Training-mode statistics of x bf16 (R, C): scale/shift/mean/rstd f32 (C) out; running_mean/var (f32 C) and
// num_batches_tracked (int64 scalar) updated in place when non-NULL (nn.BatchNorm2d training semantics: biased variance
// to normalise, unbiased into running_var, momentum as given).  partial: bq_bn_chunks(R,0,0) * 2C floats of scratch.
extern "C" __attribute__((visibility("default"))) int bq_bn_stats(const void *x, long R, int C, const float *gamma,
                                                                  const float *beta, float *running_mean,
                                                                  float *running_var, long long *num_batches_tracked,
                                                                  float eps, float momentum, float *partial,
                                                                  float *scale, float *shift, float *mean, float *rstd,
                                                                  void *stream) {
  BQ_REQUIRE(bn_extents_ok(R, C, 1, 0), BQ_ELIMIT, "bn_stats: C=%d unsupported", C);
  BQ_REQUIRE(R > 0, BQ_EINVAL, "bn_stats: empty batch");
  BQ_REQUIRE(x && gamma && beta && partial && scale && shift && mean && rstd, BQ_EINVAL, "bn_stats: null pointer");
  const int chunks = bq_bn_chunks(R, 0, 0);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(chunks), dim3(256), 0, st, (const __bf16 *)x, partial, R, C);
  BnParams p{gamma, beta, running_mean, running_var, num_batches_tracked, scale, shift, mean, rstd, eps, momentum};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, partial, (const __bf16 *)x, chunks, C, R, p);
  return check_launch("bn_stats");
}

// y = relu?(x*scale + shift): y bf16 (R, C), or with pool != 0 y bf16 (R/S, C) = max over each run of S rows; arg (pooled
// layers, may be null): u8 (R/S, C) = the row 0 .. S-1 of each group's first maximum (S <= 256).
extern "C" __attribute__((visibility("default"))) int bq_bn_apply_arg(const void *x, const float *scale, const float *shift,
                                                                      void *y, void *arg, long R, int C, int S, int relu,
                                                                      int pool, void *stream) {
  BQ_REQUIRE(bn_extents_ok(R, C, S, pool) && (C >> 3) <= 256, BQ_ELIMIT, "bn_apply: C=%d S=%d unsupported", C, S);
  BQ_REQUIRE(!arg || (pool && S <= 256), BQ_EINVAL, "bn_apply: arg is for pooled layers with S <= 256");
  if (R == 0) return BQ_OK;
  BQ_REQUIRE(x && scale && shift && y, BQ_EINVAL, "bn_apply: null pointer");
  const long rows = pool ? R / S : R, threads = rows * (C >> 3);
  const dim3 grid((unsigned)((threads + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
#define BQ_BN_APPLY(RL, PL) hipLaunchKernelGGL((bn_apply_kernel<RL, PL>), grid, dim3(256), 0, st, (const __bf16 *)x, \
                                               scale, shift, (__bf16 *)y, R, C, S, (unsigned char *)arg)
  if (relu) { if (pool) BQ_BN_APPLY(true, true); else BQ_BN_APPLY(true, false); }
  else      { if (pool) BQ_BN_APPLY(false, true); else BQ_BN_APPLY(false, false); }
#undef BQ_BN_APPLY
  return check_launch("bn_apply");
}

extern "C" __attribute__((visibility("default"))) int bq_bn_apply(const void *x, const float *scale, const float *shift,
                                                                  void *y, long R, int C, int S, int relu, int pool,
                                                                  void *stream) {
  return bq_bn_apply_arg(x, scale, shift, y, nullptr, R, C, S, relu, pool, stream);
}

// out[j] = sum over the chunks of partial[chunk][j], j < W, in chunk order (library-internal: csrc/detbwd.hip)
extern "C" int bq_bn_fold(const float *partial, float *out, int chunks, int W, void *stream) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((W + FOLD_COLS - 1) / FOLD_COLS), dim3(256), 0, (hipStream_t)stream, partial, out, chunks, W);
  return check_launch("bn_fold");
}

// The reduction half of bq_bn_backward alone: dgb f32 (2, C) = dbeta | dgamma (csrc/detbwd.hip computes the gradient w.r.t. x
// inside its fused pass).  Same arguments as bq_bn_backward without dx.
extern "C" __attribute__((visibility("default"))) int bq_bn_backward_reduce(const void *dy, const void *x, const float *scale,
                                                                            const float *shift, const float *mean,
                                                                            const float *rstd, float *partial, float *dgb, long R,
                                                                            int C, int S, int relu, int pool, void *stream) {
  BQ_REQUIRE(bn_extents_ok(R, C, S, pool) && (C >> 3) <= 256, BQ_ELIMIT, "bn_backward_reduce: C=%d S=%d unsupported", C, S);
  BQ_REQUIRE(R > 0, BQ_EINVAL, "bn_backward_reduce: empty batch");
  BQ_REQUIRE(dy && x && scale && shift && mean && rstd && partial && dgb, BQ_EINVAL, "bn_backward_reduce: null pointer");
  const int chunks = bq_bn_chunks(R, S, pool);
  hipStream_t st = (hipStream_t)stream;
#define BQ_BN_RED(RL, PL) hipLaunchKernelGGL((bn_bwd_reduce_kernel<RL, PL>), dim3(chunks), dim3(256), 0, st,           \
                                             (const __bf16 *)dy, (const __bf16 *)x, scale, shift, mean, rstd, partial, \
                                             R, C, S)
  if (relu) { if (pool) BQ_BN_RED(true, true); else BQ_BN_RED(true, false); }
  else      { if (pool) BQ_BN_RED(false, true); else BQ_BN_RED(false, false); }
#undef BQ_BN_RED
  return bq_bn_fold(partial, dgb, chunks, 2 * C, stream);
}

// Backward of bq_bn_stats + bq_bn_apply w.r.t. x, gamma, beta.  dy bf16 (R, C) or (R/S, C) when pooled;
// dgb f32 (2, C) = dbeta | dgamma (written); dx bf16 (R, C); partial: bq_bn_chunks(R,S,pool) * 2C floats.
extern "C" __attribute__((visibility("default"))) int bq_bn_backward(const void *dy, const void *x, const float *scale,
                                                                     const float *shift, const float *mean,
                                                                     const float *rstd, float *partial, float *dgb,
                                                                     void *dx, long R, int C, int S, int relu, int pool,
                                                                     void *stream) {
  BQ_REQUIRE(bn_extents_ok(R, C, S, pool) && (C >> 3) <= 256, BQ_ELIMIT, "bn_backward: C=%d S=%d unsupported", C, S);
  BQ_REQUIRE(R > 0, BQ_EINVAL, "bn_backward: empty batch");
  BQ_REQUIRE(dy && x && scale && shift && mean && rstd && partial && dgb && dx, BQ_EINVAL, "bn_backward: null pointer");
  const int chunks = bq_bn_chunks(R, S, pool);
  hipStream_t st = (hipStream_t)stream;
#define BQ_BN_RED(RL, PL) hipLaunchKernelGGL((bn_bwd_reduce_kernel<RL, PL>), dim3(chunks), dim3(256), 0, st,           \
                                             (const __bf16 *)dy, (const __bf16 *)x, scale, shift, mean, rstd, partial, \
                                             R, C, S)
  if (relu) { if (pool) BQ_BN_RED(true, true); else BQ_BN_RED(true, false); }
  else      { if (pool) BQ_BN_RED(false, true); else BQ_BN_RED(false, false); }
#undef BQ_BN_RED
  hipLaunchKernelGGL(bn_fold_kernel, dim3((2 * C + FOLD_COLS - 1) / FOLD_COLS), dim3(256), 0, st, partial, dgb, chunks, 2 * C);
  const long rows = pool ? R / S : R, threads = rows * (C >> 3);
  const dim3 grid((unsigned)((threads + 255) / 256));
#define BQ_BN_DX(RL, PL) hipLaunchKernelGGL((bn_bwd_dx_kernel<RL, PL>), grid, dim3(256), 0, st, (const __bf16 *)dy,   \
                                            (const __bf16 *)x, scale, shift, mean, rstd, dgb, (__bf16 *)dx, R, C, S)
  if (relu) { if (pool) BQ_BN_DX(true, true); else BQ_BN_DX(true, false); }
  else      { if (pool) BQ_BN_DX(false, true); else BQ_BN_DX(false, false); }
#undef BQ_BN_DX
  return check_launch("bn_backward");
}
