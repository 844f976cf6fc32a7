/*
 * oracle/pointnet2_oracle.c -- CPU restatement of the nine `pointnet2._ext` operators of
 * matthewdm0816/BridgeQA (lib/pointnet2/_ext_src).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the `cpu_baseline`
 * leg of bench.py may load it, and only as the checker / the reported CPU baseline.  The
 * product path (bridgeqa_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY PINNING: the reference ships no golden vectors or known-answer tests for these
 * operators (its only test is a CUDA-only gradcheck, lib/pointnet2/pointnet2_test.py:18-33)
 * and its kernels are CUDA-only (`AT_ASSERT(false, "CPU not supported")`,
 * _ext_src/src/sampling.cpp:34,61,83), so they cannot be executed in this image.
 * => "parity unpinned" with respect to a running reference binary.  What pins this file:
 *   (1) each function is a literal, thread-by-thread emulation of the cited CUDA kernel;
 *   (2) brute-force definitions + property tests in tests/test_oracle.py;
 *   (3) the reference's own *Python* layers imported in the build container with this library
 *       substituted for `pointnet2._ext` (oracle/gen_golden.py) -> tests/golden/ (npz files).
 *
 * Canonical arithmetic (used by this file AND by the HIP kernels; compile with
 * -ffp-contract=off): IEEE binary32, every product and sum rounded separately, evaluated in
 * the written order:  d = ((dx*dx) + (dy*dy)) + (dz*dz).  Residual risk: nvcc's default
 * -fmad=true may contract these in the shipped CUDA binary, which can flip an arg-max only on
 * sub-ulp near-ties.
 *
 * All arrays are dense row-major, float32 / int32, exactly as the reference tensors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* Squared distance.  ORACLE_FMA selects how `a*a + b*b + c*c` is rounded (a build-time variant of this same file,
 * oracle/Makefile targets libpn2oracle_fma1.so / _fma2.so; used ONLY by tests/test_oracle.py's sensitivity study of the
 * risk named above -- sampling_gpu.cu:97-107, ball_query_gpu.cu:31-33, interpolate_gpu.cu:33-35 under nvcc -fmad=true):
 *   0 (canonical)  fl(fl(fl(a*a) + fl(b*b)) + fl(c*c))        -- every product and sum rounded
 *   1              fma(c, c, fma(b, b, fl(a*a)))              -- the left-to-right chain nvcc emits for ((a*a + b*b) + c*c)
 *   2              fma(c, c, fma(a, a, fl(b*b)))              -- the other legal contraction of the inner sum */
#ifndef ORACLE_FMA
#define ORACLE_FMA 0
#endif
static inline float sq3(float a, float b, float c) {
#if ORACLE_FMA == 1
  return fmaf(c, c, fmaf(b, b, a * a));
#elif ORACLE_FMA == 2
  return fmaf(c, c, fmaf(a, a, b * b));
#else
  return (a * a) + (b * b) + (c * c);
#endif
}
ORACLE_API int oracle_fma_variant(void) { return ORACLE_FMA; }

/* include/cuda_utils.h:13-19 -- TOTAL_THREADS=512; opt_n_threads(w) = clamp(2^trunc(log(w)/log(2)), 1, 512) */
ORACLE_API int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* ------------------------------------------------------------------------------------------
 * furthest_point_sampling   sampling_gpu.cu:59-173 (kernel), :175-229 (launch: grid=b,
 * block=opt_n_threads(n)), sampling.cpp:66-87 (idx zero-init, temp filled with 1e10).
 * One "block" per batch element; block_size threads; thread tid owns points k == tid mod bs.
 * Emulated literally: per-thread strict-'>' running best in increasing k, then the
 * stride-halving LDS tree with left-wins-ties (__update, :59-65).
 * temp (b,n) is caller-provided scratch, as in the reference.
 * ---------------------------------------------------------------------------------------- */
ORACLE_API void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                               float *temp, int32_t *idxs) {
  if (m <= 0) return; /* :73 */
  const int bs = oracle_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    const float *pts = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int32_t *out = idxs + (size_t)bi * m;
    float *dists = (float *)malloc(sizeof(float) * bs);
    int *dists_i = (int *)malloc(sizeof(int) * bs);
    int old = 0;
    out[0] = old; /* :85-86 */
    for (int j = 1; j < m; ++j) {
      for (int t = 0; t < bs; ++t) { dists[t] = -1.0f; dists_i[t] = 0; } /* :90-91 */
      const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
      /* k-major sweep == per-thread tid-strided sweeps (each thread sees its k ascending) */
      for (int k = 0; k < n; ++k) {
        const int t = k % bs;
        const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
        const float mag = sq3(x2, y2, z2);
        if ((double)mag <= 1e-3) continue; /* :100-101 float promoted against a double literal */
        const float d = sq3(x2 - x1, y2 - y1, z2 - z1);
        const float d2 = d < tmp[k] ? d : tmp[k]; /* min(d, temp[k]) */
        tmp[k] = d2;
        if (d2 > dists[t]) { dists_i[t] = k; dists[t] = d2; } /* :108-109 */
      }
      for (int s = bs / 2; s >= 1; s >>= 1) { /* :115-168 */
        for (int t = 0; t < s; ++t) {
          const float v1 = dists[t], v2 = dists[t + s];
          const int i1 = dists_i[t], i2 = dists_i[t + s];
          dists[t] = v1 > v2 ? v1 : v2;
          dists_i[t] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old; /* :170-171 */
    }
    free(dists);
    free(dists_i);
  }
}

/* gather_points  sampling_gpu.cu:8-20   out[b,c,j] = points[b,c,idx[b,j]] */
ORACLE_API void oracle_gather_points(int b, int c, int n, int m, const float *points,
                                     const int32_t *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

/* gather_points_grad  sampling_gpu.cu:34-47  (grad_points zero-initialised by the caller,
 * sampling.cpp:44-46); sequential accumulation in j order. */
ORACLE_API void oracle_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                          const int32_t *idx, float *grad_points) {
#pragma omp parallel for collapse(2)
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ball_query  ball_query_gpu.cu:9-44  (idx zero-initialised by the caller, ball_query.cpp:19-21)
 * first `nsample` k (ascending) with d2 < radius*radius; first hit fills every slot. */
ORACLE_API void oracle_ball_query(int b, int n, int m, float radius, int nsample,
                                  const float *new_xyz, const float *xyz, int32_t *idx) {
  const float radius2 = radius * radius; /* :22 */
#pragma omp parallel for collapse(2) schedule(static, 16)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < m; ++j) {
      const float *p = xyz + (size_t)bi * n * 3;
      const float *q = new_xyz + ((size_t)bi * m + j) * 3;
      int32_t *o = idx + ((size_t)bi * m + j) * nsample;
      const float nx = q[0], ny = q[1], nz = q[2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
        const float d2 = sq3(nx - x, ny - y, nz - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
        }
      }
    }
}

/* group_points  group_points_gpu.cu:8-28  out[b,c,j,k] = points[b,c,idx[b,j,k]] */
ORACLE_API void oracle_group_points(int b, int c, int n, int npoints, int nsample,
                                    const float *points, const int32_t *idx, float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * npoints * nsample;
      float *o = out + ((size_t)bi * c + l) * npoints * nsample;
      for (int jk = 0; jk < npoints * nsample; ++jk) o[jk] = p[ix[jk]];
    }
}

/* group_points_grad  group_points_gpu.cu:43-64 (grad_points zero-initialised by the caller,
 * group_points.cpp:48-50).  The CUDA kernel sums with fp32 atomics in nondeterministic order;
 * this restatement sums in (j,k) order -- compare with a tolerance. */
ORACLE_API void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                         const float *grad_out, const int32_t *idx,
                                         float *grad_points) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *g = grad_points + ((size_t)bi * c + l) * n;
      const int32_t *ix = idx + (size_t)bi * npoints * nsample;
      const float *go = grad_out + ((size_t)bi * c + l) * npoints * nsample;
      for (int jk = 0; jk < npoints * nsample; ++jk) g[ix[jk]] += go[jk];
    }
}

/* three_nn  interpolate_gpu.cu:9-59  three smallest squared distances, strict '<', running
 * bests held in double initialised to 1e40 (:27), candidates are float. */
ORACLE_API void oracle_three_nn(int b, int n, int m, const float *unknown, const float *known,
                                float *dist2, int32_t *idx) {
#pragma omp parallel for collapse(2) schedule(static, 64)
  for (int bi = 0; bi < b; ++bi)
    for (int j = 0; j < n; ++j) {
      const float *u = unknown + ((size_t)bi * n + j) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        const float d = sq3(ux - x, uy - y, uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + j) * 3;
      int32_t *oi = idx + ((size_t)bi * n + j) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3; /* 1e40 -> +inf */
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

/* three_interpolate  interpolate_gpu.cu:72-101  out = p1*w1 + p2*w2 + p3*w3 (left to right) */
ORACLE_API void oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                                         const int32_t *idx, const float *weight, float *out) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * m;
      const int32_t *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j)
        o[j] = p[ix[j * 3 + 0]] * w[j * 3 + 0] + p[ix[j * 3 + 1]] * w[j * 3 + 1] +
               p[ix[j * 3 + 2]] * w[j * 3 + 2];
    }
}

/* three_interpolate_grad  interpolate_gpu.cu:116-143 (grad_points zero-initialised by the
 * caller, interpolate.cpp:85-87); atomics in the reference, (j,t) order here. */
ORACLE_API void oracle_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                              const int32_t *idx, const float *weight,
                                              float *grad_points) {
#pragma omp parallel for collapse(2)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *g = grad_points + ((size_t)bi * c + l) * m;
      const int32_t *ix = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      const float *go = grad_out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        g[ix[j * 3 + 0]] += go[j] * w[j * 3 + 0];
        g[ix[j * 3 + 1]] += go[j] * w[j * 3 + 1];
        g[ix[j * 3 + 2]] += go[j] * w[j * 3 + 2];
      }
    }
}
