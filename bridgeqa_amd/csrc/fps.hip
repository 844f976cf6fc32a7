// Furthest point sampling for gfx950 -- replaces furthest_point_sampling_kernel
// (reference lib/pointnet2/_ext_src/src/sampling_gpu.cu:69-173).
//
// Semantics reproduced exactly (see include/bqhip.h): every round picks the point maximising
// the running minimum squared distance; the reference's block-size dependent tie order is
// folded into a TOTAL ORDER key  (d desc, bitrev_{log2 bs}(k mod bs) asc, k asc)  so the result
// does not depend on how points are assigned to lanes or on the shape of the reduction.
//
// MI355X design: one workgroup per scene (rounds are strictly serial; a cross-CU hand-off per
// round costs more than the round itself).  The scene's coordinates and running minima live in
// REGISTERS for the whole kernel (PPT points per lane), the per-round arg-max is a wave64 DPP
// reduction plus one LDS exchange and ONE barrier per round; HBM is touched once (12 B/point in,
// 4 B/sample out).  Scenes too large for the register file use the streaming kernel at the bottom.
#include "bq_common.h"

namespace bq {

__device__ __forceinline__ unsigned tie_key(int k, int log2bs) {
  // smaller is better: bitrev_{log2bs}(k mod bs) in the high bits, k in the low 22
  const unsigned cls = log2bs ? (__brev((unsigned)k & ((1u << log2bs) - 1u)) >> (32 - log2bs)) : 0u;
  return (cls << 22) | (unsigned)k;
}

// Block-wide arg-max of (best, bestk) on the total order -> winning point id (uniform).  One barrier:
// each wave reduces with DPP, lane 0 folds the wave's packed (d, key) into a triple-buffered LDS slot with a
// 64-bit ds_max, everyone reads the slot back after the barrier.  `rot` cycles 0,1,2 (round j uses slot j%3;
// slot (j+2)%3 is cleared after the barrier: its readers finished before this barrier and its next writers
// cannot start before the next one).
template <int T>
__device__ __forceinline__ int block_argmax(float best, int bestk, int log2bs, unsigned long long *s_best, int rot) {
  constexpr int NW = T / 64;
  int dbits;
  unsigned wkey;
  wave_argmax(best, tie_key(bestk, log2bs), &dbits, &wkey);
  if constexpr (NW == 1) {
    return dbits < 0 ? 0 : (int)(wkey & 0x3FFFFFu);
  } else {
    if ((threadIdx.x & 63) == 0) atomicMax(&s_best[rot], pack_best(dbits, wkey));
    barrier_lds_only();
    const unsigned long long win = s_best[rot];
    if (threadIdx.x == 0) s_best[rot == 0 ? 2 : rot - 1] = 0ull;  // (rot+2)%3
    return (win >> 32) == 0ull ? 0 : (int)((~(unsigned)win) & 0x3FFFFFu);
  }
}

// ---- register-resident kernel ------------------------------------------------------------------
// T threads (multiple of bs so that a lane's points share one tie class), PPT points per lane,
// lane t owns k = t + i*T.  XYZ_LDS: also keep an LDS copy of xyz for the winner broadcast.
template <int T, int PPT, bool XYZ_LDS>
__global__ __launch_bounds__(T) void fps_reg_kernel(const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                    int N, int m, int log2bs) {
  extern __shared__ float s_xyz[];
  __shared__ unsigned long long s_best[3];
  const int t = threadIdx.x;
  const float *P = xyz + (size_t)blockIdx.x * N * 3;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  if (t < 3) s_best[t] = 0ull;

  float x[PPT], y[PPT], z[PPT], md[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * T;
    if (k < N) {
      x[i] = P[k * 3 + 0];
      y[i] = P[k * 3 + 1];
      z[i] = P[k * 3 + 2];
      // reference: `if (mag <= 1e-3) continue;` with mag promoted to double.  float(1e-3) > 1e-3,
      // so (double)mag <= 1e-3  <=>  mag < 0.001f.
      md[i] = (sqnorm(x[i], y[i], z[i]) < 0.001f) ? -1.0f : 1e10f;
    } else {
      x[i] = y[i] = z[i] = 0.0f;
      md[i] = -1.0f;  // never a candidate: fminf(-1, d>=0) stays -1 and -1 > best(-1) is false
    }
    if constexpr (XYZ_LDS) {
      if (k < N) {
        s_xyz[k * 3 + 0] = x[i];
        s_xyz[k * 3 + 1] = y[i];
        s_xyz[k * 3 + 2] = z[i];
      }
    }
  }
  IdSink sink{out, 0};
  sink.push(0, 0, m);
  __syncthreads();

  int old = 0, rot = 0;
  for (int j = 1; j < m; ++j) {
    float px, py, pz;
    if constexpr (XYZ_LDS) {
      px = s_xyz[old * 3 + 0]; py = s_xyz[old * 3 + 1]; pz = s_xyz[old * 3 + 2];
    } else {
      px = P[old * 3 + 0]; py = P[old * 3 + 1]; pz = P[old * 3 + 2];
    }
    float best = -1.0f;
    int bestk = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = sqdist(x[i], y[i], z[i], px, py, pz);
      md[i] = fminf(md[i], d);
      const bool gt = md[i] > best;
      bestk = gt ? (t + i * T) : bestk;
      best = gt ? md[i] : best;
    }
    old = block_argmax<T>(best, bestk, log2bs, s_best, rot);
    rot = rot == 2 ? 0 : rot + 1;
    sink.push(j, old, m);
  }
}

// ---- streaming kernel (any N): running minima in `temp` (global, L2-resident), like the
// reference but with the same total-order reduction.  Placeholder for scenes beyond the
// register file until the bucketed kernel lands.
template <int T>
__global__ __launch_bounds__(T) void fps_stream_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                       int32_t *__restrict__ idx, int N, int m, int log2bs) {
  __shared__ unsigned long long s_best[3];
  const int t = threadIdx.x;
  const float *P = xyz + (size_t)blockIdx.x * N * 3;
  float *md = temp + (size_t)blockIdx.x * N;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  if (t < 3) s_best[t] = 0ull;
  for (int k = t; k < N; k += T) md[k] = (sqnorm(P[k * 3], P[k * 3 + 1], P[k * 3 + 2]) < 0.001f) ? -1.0f : 1e10f;
  IdSink sink{out, 0};
  sink.push(0, 0, m);
  __syncthreads();
  int old = 0, rot = 0;
  for (int j = 1; j < m; ++j) {
    const float px = P[old * 3 + 0], py = P[old * 3 + 1], pz = P[old * 3 + 2];
    float best = -1.0f;
    int bestk = 0;
    for (int k = t; k < N; k += T) {  // each lane revisits only its own k: no cross-lane hazard on md
      const float d = sqdist(P[k * 3 + 0], P[k * 3 + 1], P[k * 3 + 2], px, py, pz);
      const float v = fminf(md[k], d);
      md[k] = v;
      const bool gt = v > best;
      bestk = gt ? k : bestk;
      best = gt ? v : best;
    }
    old = block_argmax<T>(best, bestk, log2bs, s_best, rot);
    rot = rot == 2 ? 0 : rot + 1;
    sink.push(j, old, m);
  }
}

template <int T, int PPT>
static void launch_reg(const float *xyz, int32_t *idx, int B, int N, int m, int log2bs, hipStream_t st) {
  const size_t lds_bytes = (size_t)N * 12;
  if (lds_bytes <= 96 * 1024) {
    hipLaunchKernelGGL((fps_reg_kernel<T, PPT, true>), dim3(B), dim3(T), lds_bytes, st, xyz, idx, N, m, log2bs);
  } else {
    hipLaunchKernelGGL((fps_reg_kernel<T, PPT, false>), dim3(B), dim3(T), 0, st, xyz, idx, N, m, log2bs);
  }
}

}  // namespace bq

extern "C" int bq_opt_n_threads(int work_size) {
  // cuda_utils.h:15-19, evaluated in double exactly as written there
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

namespace bq {
int launch_fps_bucket(const float *xyz, void *workspace, size_t workspace_bytes, int32_t *idx, int B, int N, int m,
                      int log2bs, hipStream_t st);
}

extern "C" int bq_furthest_point_sampling(const float *xyz, void *workspace, size_t workspace_bytes, int32_t *idx,
                                          int B, int N, int m, void *stream) {
  using namespace bq;
  BQ_REQUIRE(B >= 0 && N >= 1 && m >= 0, BQ_EINVAL, "fps: bad extents B=%d N=%d m=%d", B, N, m);
  BQ_REQUIRE(N < (1 << 22), BQ_ELIMIT, "fps: N=%d exceeds 2^22-1", N);
  if (B == 0 || m == 0) return BQ_OK;  // sampling_gpu.cu:73
  BQ_REQUIRE(xyz && idx, BQ_EINVAL, "fps: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int bs = bq_opt_n_threads(N);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  // T must be a multiple of bs (bs <= 256 when N < 512, else 512)
  if (N <= 256)        launch_reg<256, 1>(xyz, idx, B, N, m, log2bs, st);
  else if (N < 512)    launch_reg<256, 2>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 512)   launch_reg<512, 1>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 1024)  launch_reg<512, 2>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 2048)  launch_reg<512, 4>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 4096)  launch_reg<1024, 4>(xyz, idx, B, N, m, log2bs, st);
  else return launch_fps_bucket(xyz, workspace, workspace_bytes, idx, B, N, m, log2bs, st);
  return check_launch("furthest_point_sampling");
}

// Brute-force (unpruned) kernels, any N: for A/B timing against the bucketed kernel and as an independent on-GPU
// cross-check of it.  temp: (B,N) f32 scratch, needed when N > 24576.
extern "C" int bq_furthest_point_sampling_bruteforce(const float *xyz, float *temp, int32_t *idx, int B, int N,
                                                     int m, void *stream) {
  using namespace bq;
  BQ_REQUIRE(B >= 0 && N >= 1 && m >= 0, BQ_EINVAL, "fps: bad extents B=%d N=%d m=%d", B, N, m);
  BQ_REQUIRE(N < (1 << 22), BQ_ELIMIT, "fps: N=%d exceeds 2^22-1", N);
  if (B == 0 || m == 0) return BQ_OK;
  BQ_REQUIRE(xyz && idx, BQ_EINVAL, "fps: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int bs = bq_opt_n_threads(N);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  if (N <= 4096) return bq_furthest_point_sampling(xyz, nullptr, 0, idx, B, N, m, stream);
  if (N <= 8192)       launch_reg<1024, 8>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 16384) launch_reg<1024, 16>(xyz, idx, B, N, m, log2bs, st);
  else if (N <= 24576) launch_reg<1024, 24>(xyz, idx, B, N, m, log2bs, st);
  else {
    BQ_REQUIRE(temp, BQ_EINVAL, "fps: temp scratch required for N=%d", N);
    hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(B), dim3(1024), 0, st, xyz, temp, idx, N, m, log2bs);
  }
  return check_launch("furthest_point_sampling(bruteforce)");
}
