// Ball query, grouping, gathering, three-NN and three-interpolate for gfx950 -- replaces
// ball_query_gpu.cu, group_points_gpu.cu, interpolate_gpu.cu and the gather kernels of
// sampling_gpu.cu of the reference (lib/pointnet2/_ext_src/src).  The reference launches
// gridDim.x = B blocks for every op (16 blocks on a 256-CU part); here every op is decomposed
// over (scene, centre/row tiles) so a launch fills the chip, and output rows are written with
// 16-byte stores.
#include <stdarg.h>
#include <string.h>

#include "bq_common.h"

namespace bq {

static thread_local char g_err[256] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cus() {
  // per device ordinal (a function-static of the FIRST device would size every later device's grids with its count)
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cached[dev] = cus > 0 ? cus : 256;
  }
  return cached[dev];
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// ball query: one wave per PAIR of centres.  The wave sweeps the scene 64 points at a time; a ballot of
// the in-radius lanes plus a prefix popcount gives each hit its slot in ascending-id order, which
// is exactly the order the reference's serial scan (ball_query_gpu.cu:28-42) produces.
// A ball with fewer than S points scans the whole scene (625 chunks at N = 40000: the common case at SA1's radius), so
// the kernel is a stream of the scene's coordinates through L2 per centre -- 15.7 GB per SA1 launch at one centre per
// wave, 21.7 TB/s.  Hence (round 3): four chunks per trip with all loads issued before the first comparison (the
// one-chunk loop was a chain of exposed L2 latencies: 1270 -> 720 us), and TWO centres per wave on the same loaded
// points (half the traffic).  Chunks are consumed in order per centre, so slots, padding and the early exit are exactly
// the serial scan's.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ball_query_wave_kernel(const float *__restrict__ new_xyz,
                                                              const float *__restrict__ xyz,
                                                              int32_t *__restrict__ idx, int N, int M, float radius2,
                                                              int S) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const float *P = xyz + (size_t)b * N * 3;
  // (grid-stride over the centre pairs: bq_ball_query_background launches about one workgroup per CU, so that a large
  // query running beside latency-bound kernels of another stream does not hold every wave slot of the chip)
  for (int j0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2; j0 < M; j0 += gridDim.x * 8) {
  const bool two = j0 + 1 < M;   // (wave-uniform)
  const float *q = new_xyz + ((size_t)b * M + j0) * 3;
  int32_t *o0 = idx + ((size_t)b * M + j0) * S, *o1 = o0 + S;
  const float qx0 = q[0], qy0 = q[1], qz0 = q[2];
  const float qx1 = two ? q[3] : qx0, qy1 = two ? q[4] : qy0, qz1 = two ? q[5] : qz0;
  int cnt0 = 0, first0 = 0, cnt1 = two ? 0 : S, first1 = 0;
  constexpr int U = 4;
  for (int base = 0; base < N && (cnt0 < S || cnt1 < S); base += 64 * U) {
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(base + 64 * u + lane, N - 1);
      px[u] = P[k * 3 + 0]; py[u] = P[k * 3 + 1]; pz[u] = P[k * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = base + 64 * u + lane;
      const bool in = k < N;
      const bool hit0 = in && cnt0 < S && sqdist(qx0, qy0, qz0, px[u], py[u], pz[u]) < radius2;
      const bool hit1 = in && cnt1 < S && sqdist(qx1, qy1, qz1, px[u], py[u], pz[u]) < radius2;
      const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
      if (m0) {
        if (cnt0 == 0) first0 = base + 64 * u + __builtin_ctzll(m0);
        const int pos = cnt0 + __builtin_popcountll(m0 & ((1ull << lane) - 1ull));
        if (hit0 && pos < S) o0[pos] = k;
        cnt0 += __builtin_popcountll(m0);
      }
      if (m1) {
        if (cnt1 == 0) first1 = base + 64 * u + __builtin_ctzll(m1);
        const int pos = cnt1 + __builtin_popcountll(m1 & ((1ull << lane) - 1ull));
        if (hit1 && pos < S) o1[pos] = k;
        cnt1 += __builtin_popcountll(m1);
      }
    }
  }
  if (cnt0 > S) cnt0 = S;
  if (cnt1 > S) cnt1 = S;
  for (int s = cnt0 + lane; s < S; s += 64) o0[s] = first0;  // pad with the first hit (0 if none)
  if (two)
    for (int s = cnt1 + lane; s < S; s += 64) o1[s] = first1;
  }
}

// ---------------------------------------------------------------------------------------------
// gather_points:  out[b,c,j] = points[b,c,idx[b,j]]
// ---------------------------------------------------------------------------------------------
__global__ void gather_points_kernel(const float *__restrict__ points, const int32_t *__restrict__ idx,
                                     float *__restrict__ out, int C, int N, int M) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  out[((size_t)b * C + c) * M + j] = points[((size_t)b * C + c) * N + idx[(size_t)b * M + j]];
}

__global__ void gather_points_grad_kernel(const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                          float *__restrict__ grad_points, int C, int N, int M) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  atomicAdd(grad_points + ((size_t)b * C + c) * N + idx[(size_t)b * M + j], grad_out[((size_t)b * C + c) * M + j]);
}

// ---------------------------------------------------------------------------------------------
// group_points:  out[b,c,j,k] = points[b,c,idx[b,j,k]].  Flat index over (j,k); a thread owns VEC
// consecutive outputs of one channel row so the (dominant) write stream is 16 B per lane; the
// gathered reads come from one channel row (N*4 bytes) which stays in L2.
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx, float *__restrict__ out,
                                                           int C, int N, int MS) {
  const int b = blockIdx.z;
  const int e = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (e >= MS) return;
  const int32_t *ix = idx + (size_t)b * MS + e;
  int id[VEC];
  if constexpr (VEC == 4) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
  } else {
    id[0] = ix[0];
  }
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float *p = points + ((size_t)b * C + c) * N;
    float *o = out + ((size_t)b * C + c) * MS + e;
    if constexpr (VEC == 4) {
      float4 v;
      v.x = p[id[0]]; v.y = p[id[1]]; v.z = p[id[2]]; v.w = p[id[3]];
      *reinterpret_cast<float4 *>(o) = v;
    } else {
      o[0] = p[id[0]];
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void group_points_grad_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ grad_points, int C, int N, int MS) {
  const int b = blockIdx.z;
  const int e = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (e >= MS) return;
  const int32_t *ix = idx + (size_t)b * MS + e;
  int id[VEC];
  if constexpr (VEC == 4) {
    const int4 v = *reinterpret_cast<const int4 *>(ix);
    id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
  } else {
    id[0] = ix[0];
  }
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float *g = grad_points + ((size_t)b * C + c) * N;
    const float *go = grad_out + ((size_t)b * C + c) * MS + e;
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4 *>(go);
      // padded neighbourhoods repeat one id: pre-sum runs of equal ids to cut atomic traffic
      float acc = v.x;
      if (id[1] == id[0]) acc += v.y; else { atomicAdd(g + id[0], acc); acc = v.y; }
      if (id[2] == id[1]) acc += v.z; else { atomicAdd(g + id[1], acc); acc = v.z; }
      if (id[3] == id[2]) acc += v.w; else { atomicAdd(g + id[2], acc); acc = v.w; }
      atomicAdd(g + id[3], acc);
    } else {
      atomicAdd(g + id[0], go[0]);
    }
  }
}

typedef __bf16 bq_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void store4(__bf16 *p, float4 v) {
  bq_bf16x4 w;
  w[0] = (__bf16)v.x; w[1] = (__bf16)v.y; w[2] = (__bf16)v.z; w[3] = (__bf16)v.w;
  *reinterpret_cast<bq_bf16x4 *>(p) = w;
}
__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __bf16 *p) {
  const bq_bf16x4 w = *reinterpret_cast<const bq_bf16x4 *>(p);
  return make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
}

// ---------------------------------------------------------------------------------------------
// fused QueryAndGroup tail (pointnet2_utils.py:348-359): rows 0..2 = (xyz[idx]-centre)/radius,
// rows 3.. = features[idx]; written once, straight into the SharedMLP's input layout.
// ---------------------------------------------------------------------------------------------
template <typename OT>  // OT = float (reference precision) or __bf16 (input of the bf16 SharedMLP GEMMs)
__global__ __launch_bounds__(256) void group_concat_kernel(const float *__restrict__ xyz,
                                                           const float *__restrict__ new_xyz,
                                                           const float *__restrict__ features,
                                                           const int32_t *__restrict__ idx, OT *__restrict__ out,
                                                           int C, int N, int M, int S, float radius, int normalize) {
  const int b = blockIdx.z;
  const int MS = M * S;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;  // S % 4 == 0 => the 4 outputs share a centre
  if (e >= MS) return;
  const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)b * MS + e);
  const int CT = C + 3;
  if (blockIdx.y == 0) {
    const int j = e / S;
    const float *P = xyz + (size_t)b * N * 3;
    const float *q = new_xyz + ((size_t)b * M + j) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float qc = q[c];
      float4 v;
      v.x = P[id.x * 3 + c] - qc; v.y = P[id.y * 3 + c] - qc; v.z = P[id.z * 3 + c] - qc; v.w = P[id.w * 3 + c] - qc;
      if (normalize) { v.x /= radius; v.y /= radius; v.z /= radius; v.w /= radius; }
      store4(out + ((size_t)b * CT + c) * MS + e, v);
    }
  }
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float *p = features + ((size_t)b * C + c) * N;
    float4 v;
    v.x = p[id.x]; v.y = p[id.y]; v.z = p[id.z]; v.w = p[id.w];
    store4(out + ((size_t)b * CT + c + 3) * MS + e, v);
  }
}

template <typename OT>
__global__ __launch_bounds__(256) void group_concat_grad_kernel(const OT *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ grad_features,
                                                                float *__restrict__ grad_xyz,
                                                                float *__restrict__ grad_new_xyz, int C, int N, int M,
                                                                int S, float radius, int normalize) {
  const int b = blockIdx.z;
  const int MS = M * S;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= MS) return;
  const int4 idv = *reinterpret_cast<const int4 *>(idx + (size_t)b * MS + e);
  const int id[4] = {idv.x, idv.y, idv.z, idv.w};
  const int CT = C + 3;
  if (blockIdx.y == 0 && (grad_xyz || grad_new_xyz)) {
    const int j = e / S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 v = load4(grad_out + ((size_t)b * CT + c) * MS + e);
      if (normalize) { v.x /= radius; v.y /= radius; v.z /= radius; v.w /= radius; }
      const float g[4] = {v.x, v.y, v.z, v.w};
      if (grad_xyz) {
#pragma unroll
        for (int t = 0; t < 4; ++t) atomicAdd(grad_xyz + ((size_t)b * N + id[t]) * 3 + c, g[t]);
      }
      if (grad_new_xyz) atomicAdd(grad_new_xyz + ((size_t)b * M + j) * 3 + c, -((g[0] + g[1]) + (g[2] + g[3])));
    }
  }
  if (!grad_features) return;
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float *g = grad_features + ((size_t)b * C + c) * N;
    const float4 v = load4(grad_out + ((size_t)b * CT + c + 3) * MS + e);
    float acc = v.x;
    if (id[1] == id[0]) acc += v.y; else { atomicAdd(g + id[0], acc); acc = v.y; }
    if (id[2] == id[1]) acc += v.z; else { atomicAdd(g + id[1], acc); acc = v.z; }
    if (id[3] == id[2]) acc += v.w; else { atomicAdd(g + id[2], acc); acc = v.w; }
    atomicAdd(g + id[3], acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Point-major ("channels-last") grouping: out[b][j][k][0:3] = (xyz[idx]-centre)/radius, out[b][j][k][3:3+C] =
// feats[b][idx][:].  With features stored point-major (B,N,C) a neighbourhood gather is a copy of contiguous
// rows: one wave per output position reads the point's row and writes the output row, both coalesced.  The
// output feeds the SharedMLP as an NHWC tensor (no layout transposes inside the convolution library).
// ---------------------------------------------------------------------------------------------
template <typename OT>
__global__ __launch_bounds__(256) void group_concat_pm_kernel(const float *__restrict__ xyz,
                                                              const float *__restrict__ new_xyz,
                                                              const float *__restrict__ feats, long f_bs, long f_rs,
                                                              const int32_t *__restrict__ idx, OT *__restrict__ out,
                                                              int C, int N, int M, int S, float radius, int normalize,
                                                              long total, int ld) {
  const int lane = threadIdx.x & 63;
  const int CT = C + 3;
  for (long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pos < total; pos += (long)gridDim.x * 4) {
    const long bj = pos / S;            // b*M + j
    const int b = (int)(bj / M);
    const int id = idx[pos];
    OT *o = out + pos * ld;             // rows of ld >= 3 + C elements (padding zeroed: 16-byte aligned GEMM rows)
    if (lane < ld - CT) o[CT + lane] = (OT)0.f;
    if (lane < 3) {
      float v = xyz[((long)b * N + id) * 3 + lane] - new_xyz[bj * 3 + lane];
      if (normalize) v /= radius;
      o[lane] = (OT)v;
    }
    const float *f = feats + (long)b * f_bs + (long)id * f_rs;
    for (int c = lane; c < C; c += 64) o[3 + c] = (OT)f[c];
  }
}

// The same for bf16 rows with 16-byte accesses (round 5).  The kernel above moves one ELEMENT per lane -- 2-byte stores, 4-byte
// loads: three load and five store instructions per 272-byte row, each a full 64-lane pass through the address unit, ~130 cycles
// of it per row and 0.45 ms for SA1's 2.1 M rows (571 MB: a fifth of the HBM rate).  Here a wave assembles RW = 8 consecutive
// rows in its own LDS strip -- one 16-byte (dword-aligned) load per lane and row for the features, one load instruction for the
// eight rows' coordinates and centres, one for their indices -- and writes the strip out as whole 16-byte chunks of one
// contiguous 8 x ld x 2-byte region: 13 vector-memory instructions per 8 rows instead of ~70.
// Requires C % 4 == 0, C <= 256, ld <= 264, S % 8 == 0 (a strip never straddles two centres), total % 8 == 0.
struct __attribute__((packed, aligned(4))) f4u_t { float v[4]; };
constexpr int GC_RW = 8, GC_ROWB = 528;
__global__ __launch_bounds__(256) void group_concat_pm_vec_kernel(const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                  const float *__restrict__ feats, long f_bs, long f_rs,
                                                                  const int32_t *__restrict__ idx, __bf16 *__restrict__ out,
                                                                  int C, int N, int M, int S, float radius, int normalize,
                                                                  long total, int ld) {
  __shared__ __attribute__((aligned(16))) unsigned char s_buf[4][GC_RW * GC_ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char *sb = s_buf[wave];
  const int CT = C + 3, rowb = ld * 2, nch = GC_RW * rowb / 16;
  const int r6 = lane / 6, a6 = lane - r6 * 6;
  for (long pos0 = ((long)blockIdx.x * 4 + wave) * GC_RW; pos0 < total; pos0 += (long)gridDim.x * 4 * GC_RW) {
    const long bj = pos0 / S;             // b * M + j: one centre for the whole strip
    const int b = (int)(bj / M);
    const int myid = lane < GC_RW ? idx[pos0 + lane] : 0;
    const int id6 = __shfl(myid, r6 < GC_RW ? r6 : 0);
    float xv = 0.f;
    if (lane < GC_RW * 6) xv = a6 < 3 ? xyz[((long)b * N + id6) * 3 + a6] : new_xyz[bj * 3 + (a6 - 3)];
    f4u_t fv[GC_RW];
    const float *fb = feats + (long)b * f_bs + lane * 4;
#pragma unroll
    for (int r = 0; r < GC_RW; ++r) {
      const int id = __builtin_amdgcn_readlane(myid, r);
      if (lane * 4 < C) fv[r] = *reinterpret_cast<const f4u_t *>(fb + (long)id * f_rs);
    }
    {
      const float ctr = __shfl(xv, lane + 3);
      float d = xv - ctr;
      if (normalize) d /= radius;
      if (lane < GC_RW * 6 && a6 < 3) *reinterpret_cast<__bf16 *>(sb + r6 * rowb + a6 * 2) = (__bf16)d;
    }
#pragma unroll
    for (int r = 0; r < GC_RW; ++r) {
      if (lane * 4 < C) {
        __bf16 *o = reinterpret_cast<__bf16 *>(sb + r * rowb) + 3 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)fv[r].v[e];
      }
      if (lane < ld - CT) reinterpret_cast<__bf16 *>(sb + r * rowb)[CT + lane] = (__bf16)0.f;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned char *dst = reinterpret_cast<unsigned char *>(out + pos0 * ld);
    for (int q = lane; q < nch; q += 64)
      *reinterpret_cast<uint4 *>(dst + q * 16) = *reinterpret_cast<const uint4 *>(sb + q * 16);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// grad of the above w.r.t. point-major features (B,N,C) f32 (zero_init) and xyz / new_xyz (zero_init, optional)
template <typename OT>
__global__ __launch_bounds__(256) void group_concat_pm_grad_kernel(const OT *__restrict__ grad_out,
                                                                   const int32_t *__restrict__ idx,
                                                                   float *__restrict__ grad_feats,
                                                                   float *__restrict__ grad_xyz,
                                                                   float *__restrict__ grad_new_xyz, int C, int N,
                                                                   int M, int S, float radius, int normalize,
                                                                   long total, int ld) {
  const int lane = threadIdx.x & 63;
  for (long pos = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pos < total; pos += (long)gridDim.x * 4) {
    const long bj = pos / S;
    const int b = (int)(bj / M);
    const int id = idx[pos];
    const OT *g = grad_out + pos * ld;
    if (lane < 3 && (grad_xyz || grad_new_xyz)) {
      float v = (float)g[lane];
      if (normalize) v /= radius;
      if (grad_xyz) atomicAdd(grad_xyz + ((long)b * N + id) * 3 + lane, v);
      if (grad_new_xyz) atomicAdd(grad_new_xyz + bj * 3 + lane, -v);
    }
    if (grad_feats) {
      float *gf = grad_feats + ((long)b * N + id) * C;
      for (int c = lane; c < C; c += 64) atomicAdd(gf + c, (float)g[3 + c]);  // 256 contiguous bytes per wave-instruction
    }
  }
}

// ---------------------------------------------------------------------------------------------
// three_nn: one lane per unknown point, known points staged through LDS in tiles and read as
// wave-wide broadcasts; strict '<' with ascending k keeps the lowest index on ties, and the
// running bests are compared exactly as the reference does (float candidate against a best
// that is either a float value or the 1e40 sentinel -- held here as +inf, same ordering).
// ---------------------------------------------------------------------------------------------
template <bool SQRT>
__global__ __launch_bounds__(256) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx, int n, int m) {
  __shared__ float s_k[1024 * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const float *K = known + (size_t)b * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) {
    const float *u = unknown + ((size_t)b * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  const float INF = __int_as_float(0x7f800000);
  float b1 = INF, b2 = INF, b3 = INF;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int base = 0; base < m; base += 1024) {
    const int cnt = min(1024, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt * 3; t += 256) s_k[t] = K[base * 3 + t];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const float d = sqdist(ux, uy, uz, s_k[k * 3 + 0], s_k[k * 3 + 1], s_k[k * 3 + 2]);
      const int kk = base + k;
      if (d < b1) {
        b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = kk;
      } else if (d < b2) {
        b3 = b2; i3 = i2; b2 = d; i2 = kk;
      } else if (d < b3) {
        b3 = d; i3 = kk;
      }
    }
  }
  if (j < n) {
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int32_t *oi = idx + ((size_t)b * n + j) * 3;
    if constexpr (SQRT) {  // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt)
      b1 = sqrtf(b1); b2 = sqrtf(b2); b3 = sqrtf(b3);
    }
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_kernel(const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out, int C, int m, int n) {
#pragma clang fp contract(off)
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int32_t *ix = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    const float *p = points + ((size_t)b * C + c) * m;
    const float a = p[i1] * w1;
    const float bb = p[i2] * w2;
    const float cc = p[i3] * w3;
    out[((size_t)b * C + c) * n + j] = (a + bb) + cc;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(const float *__restrict__ grad_out,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ grad_points, int C, int n,
                                                                     int m) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int32_t *ix = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  for (int c = blockIdx.y; c < C; c += gridDim.y) {
    float *g = grad_points + ((size_t)b * C + c) * m;
    const float go = grad_out[((size_t)b * C + c) * n + j];
    atomicAdd(g + i1, go * w1);
    atomicAdd(g + i2, go * w2);
    atomicAdd(g + i3, go * w3);
  }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// channel-dimension grid size: enough blocks to fill the chip, the rest strided inside the kernel
static inline int cgrid(int C, int xblocks, int B) {
  long want = (2048 + (long)xblocks * B - 1) / ((long)xblocks * B);
  if (want < 1) want = 1;
  if (want > C) want = C;
  return (int)(want < 1 ? 1 : want);
}

}  // namespace bq

using namespace bq;

extern "C" int bq_abi_version(void) { return BQHIP_ABI_VERSION; }
extern "C" const char *bq_last_error(void) { return bq::g_err; }

static int ball_query_launch(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                             int nsample, bool background, void *stream) {
  BQ_REQUIRE(B >= 0 && N >= 0 && M >= 0 && nsample >= 0, BQ_EINVAL, "ball_query: bad extents");
  if (B == 0 || M == 0 || nsample == 0) return BQ_OK;
  BQ_REQUIRE(new_xyz && idx && (xyz || N == 0), BQ_EINVAL, "ball_query: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "ball_query: B=%d > 65535", B);
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22, rounded to fp32 on the host
  int gx = cdiv(M, 8);   // a workgroup = 4 waves = 8 centres
  if (background) {      // about one workgroup per CU over the whole batch (bq_ball_query_background, include/bqhip.h)
    const int cus = device_cus();
    const int per_scene = cus / B > 0 ? cus / B : 1;
    if (per_scene < gx) gx = per_scene;
  }
  hipLaunchKernelGGL(ball_query_wave_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, new_xyz, xyz,
                     idx, N, M, radius2, nsample);
  return check_launch("ball_query");
}

extern "C" int bq_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                             int nsample, void *stream) {
  return ball_query_launch(new_xyz, xyz, idx, B, N, M, radius, nsample, false, stream);
}

extern "C" int bq_ball_query_background(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M,
                                        float radius, int nsample, void *stream) {
  return ball_query_launch(new_xyz, xyz, idx, B, N, M, radius, nsample, true, stream);
}

extern "C" int bq_gather_points(const float *points, const int32_t *idx, float *out, int B, int C, int N, int M,
                                void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, BQ_EINVAL, "gather_points: bad extents");
  if (B == 0 || C == 0 || M == 0) return BQ_OK;
  BQ_REQUIRE(points && idx && out, BQ_EINVAL, "gather_points: null pointer");
  BQ_REQUIRE(B <= 65535 && C <= 65535, BQ_ELIMIT, "gather_points: B or C > 65535");
  hipLaunchKernelGGL(gather_points_kernel, dim3(cdiv(M, 256), C, B), dim3(256), 0, (hipStream_t)stream, points, idx,
                     out, C, N, M);
  return check_launch("gather_points");
}

extern "C" int bq_gather_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C,
                                     int N, int M, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0, BQ_EINVAL, "gather_points_grad: bad extents");
  if (B == 0 || C == 0 || M == 0) return BQ_OK;
  BQ_REQUIRE(grad_out && idx && grad_points, BQ_EINVAL, "gather_points_grad: null pointer");
  BQ_REQUIRE(B <= 65535 && C <= 65535, BQ_ELIMIT, "gather_points_grad: B or C > 65535");
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(cdiv(M, 256), C, B), dim3(256), 0, (hipStream_t)stream,
                     grad_out, idx, grad_points, C, N, M);
  return check_launch("gather_points_grad");
}

extern "C" int bq_group_points(const float *points, const int32_t *idx, float *out, int B, int C, int N, int M, int S,
                               void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_points: bad extents");
  if (B == 0 || C == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(points && idx && out, BQ_EINVAL, "group_points: null pointer");
  BQ_REQUIRE(B <= 65535 && (long)M * S < (1L << 31), BQ_ELIMIT, "group_points: extent too large");
  const int MS = M * S;
  if (MS % 4 == 0) {
    const int xb = cdiv(MS / 4, 256);
    hipLaunchKernelGGL(group_points_kernel<4>, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                       points, idx, out, C, N, MS);
  } else {
    const int xb = cdiv(MS, 256);
    hipLaunchKernelGGL(group_points_kernel<1>, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                       points, idx, out, C, N, MS);
  }
  return check_launch("group_points");
}

extern "C" int bq_group_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int B, int C,
                                    int N, int M, int S, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_points_grad: bad extents");
  if (B == 0 || C == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(grad_out && idx && grad_points, BQ_EINVAL, "group_points_grad: null pointer");
  BQ_REQUIRE(B <= 65535 && (long)M * S < (1L << 31), BQ_ELIMIT, "group_points_grad: extent too large");
  const int MS = M * S;
  if (MS % 4 == 0) {
    const int xb = cdiv(MS / 4, 256);
    hipLaunchKernelGGL(group_points_grad_kernel<4>, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                       grad_out, idx, grad_points, C, N, MS);
  } else {
    const int xb = cdiv(MS, 256);
    hipLaunchKernelGGL(group_points_grad_kernel<1>, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                       grad_out, idx, grad_points, C, N, MS);
  }
  return check_launch("group_points_grad");
}

static int group_concat_impl(const float *xyz, const float *new_xyz, const float *features, const int32_t *idx,
                             void *out, int out_bf16, int B, int C, int N, int M, int S, float radius, int normalize,
                             void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_concat: bad extents");
  if (B == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(xyz && new_xyz && idx && out && (features || C == 0), BQ_EINVAL, "group_concat: null pointer");
  BQ_REQUIRE(S % 4 == 0, BQ_ELIMIT, "group_concat: nsample=%d must be a multiple of 4", S);
  BQ_REQUIRE(B <= 65535 && (long)M * S < (1L << 31), BQ_ELIMIT, "group_concat: extent too large");
  const int xb = cdiv(M * S / 4, 256);
  const int yb = C ? cgrid(C, xb, B) : 1;
  if (out_bf16)
    hipLaunchKernelGGL(group_concat_kernel<__bf16>, dim3(xb, yb, B), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz,
                       features, idx, (__bf16 *)out, C, N, M, S, radius, normalize);
  else
    hipLaunchKernelGGL(group_concat_kernel<float>, dim3(xb, yb, B), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz,
                       features, idx, (float *)out, C, N, M, S, radius, normalize);
  return check_launch("group_concat");
}

extern "C" int bq_group_concat(const float *xyz, const float *new_xyz, const float *features, const int32_t *idx,
                               float *out, int B, int C, int N, int M, int S, float radius, int normalize,
                               void *stream) {
  return group_concat_impl(xyz, new_xyz, features, idx, out, 0, B, C, N, M, S, radius, normalize, stream);
}

// same, writing bf16 (the grouped tensor is the largest activation of the detector: 1.1 GB in fp32 at SA1)
extern "C" __attribute__((visibility("default"))) int bq_group_concat_bf16(
    const float *xyz, const float *new_xyz, const float *features, const int32_t *idx, void *out, int B, int C, int N,
    int M, int S, float radius, int normalize, void *stream) {
  return group_concat_impl(xyz, new_xyz, features, idx, out, 1, B, C, N, M, S, radius, normalize, stream);
}

static int group_concat_grad_impl(const void *grad_out, int in_bf16, const int32_t *idx, float *grad_features,
                                  float *grad_xyz, float *grad_new_xyz, int B, int C, int N, int M, int S,
                                  float radius, int normalize, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_concat_grad: bad extents");
  if (B == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(grad_out && idx, BQ_EINVAL, "group_concat_grad: null pointer");
  BQ_REQUIRE(S % 4 == 0, BQ_ELIMIT, "group_concat_grad: nsample=%d must be a multiple of 4", S);
  BQ_REQUIRE(B <= 65535 && (long)M * S < (1L << 31), BQ_ELIMIT, "group_concat_grad: extent too large");
  const int xb = cdiv(M * S / 4, 256);
  const int yb = (C && grad_features) ? cgrid(C, xb, B) : 1;
  if (in_bf16)
    hipLaunchKernelGGL(group_concat_grad_kernel<__bf16>, dim3(xb, yb, B), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)grad_out, idx, grad_features, grad_xyz, grad_new_xyz, C, N, M, S, radius,
                       normalize);
  else
    hipLaunchKernelGGL(group_concat_grad_kernel<float>, dim3(xb, yb, B), dim3(256), 0, (hipStream_t)stream,
                       (const float *)grad_out, idx, grad_features, grad_xyz, grad_new_xyz, C, N, M, S, radius,
                       normalize);
  return check_launch("group_concat_grad");
}

extern "C" int bq_group_concat_grad(const float *grad_out, const int32_t *idx, float *grad_features, float *grad_xyz,
                                    float *grad_new_xyz, int B, int C, int N, int M, int S, float radius,
                                    int normalize, void *stream) {
  return group_concat_grad_impl(grad_out, 0, idx, grad_features, grad_xyz, grad_new_xyz, B, C, N, M, S, radius,
                                normalize, stream);
}

extern "C" __attribute__((visibility("default"))) int bq_group_concat_grad_bf16(
    const void *grad_out, const int32_t *idx, float *grad_features, float *grad_xyz, float *grad_new_xyz, int B, int C,
    int N, int M, int S, float radius, int normalize, void *stream) {
  return group_concat_grad_impl(grad_out, 1, idx, grad_features, grad_xyz, grad_new_xyz, B, C, N, M, S, radius,
                                normalize, stream);
}

extern "C" int bq_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx, int B, int n, int m,
                           void *stream) {
  BQ_REQUIRE(B >= 0 && n >= 0 && m >= 0, BQ_EINVAL, "three_nn: bad extents");
  if (B == 0 || n == 0) return BQ_OK;
  BQ_REQUIRE(unknown && dist2 && idx && (known || m == 0), BQ_EINVAL, "three_nn: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "three_nn: B > 65535");
  hipLaunchKernelGGL(three_nn_kernel<false>, dim3(cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, unknown, known,
                     dist2, idx, n, m);
  return check_launch("three_nn");
}

extern "C" int bq_three_nn_dist(const float *unknown, const float *known, float *dist, int32_t *idx, int B, int n,
                                int m, void *stream) {
  BQ_REQUIRE(B >= 0 && n >= 0 && m >= 0, BQ_EINVAL, "three_nn_dist: bad extents");
  if (B == 0 || n == 0) return BQ_OK;
  BQ_REQUIRE(unknown && dist && idx && (known || m == 0), BQ_EINVAL, "three_nn_dist: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "three_nn_dist: B > 65535");
  hipLaunchKernelGGL(three_nn_kernel<true>, dim3(cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, unknown, known,
                     dist, idx, n, m);
  return check_launch("three_nn_dist");
}

extern "C" int bq_three_interpolate(const float *points, const int32_t *idx, const float *weight, float *out, int B,
                                    int C, int m, int n, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && n >= 0 && m >= 0, BQ_EINVAL, "three_interpolate: bad extents");
  if (B == 0 || C == 0 || n == 0) return BQ_OK;
  BQ_REQUIRE(points && idx && weight && out, BQ_EINVAL, "three_interpolate: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "three_interpolate: B > 65535");
  const int xb = cdiv(n, 256);
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                     points, idx, weight, out, C, m, n);
  return check_launch("three_interpolate");
}

extern "C" int bq_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                                         float *grad_points, int B, int C, int n, int m, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && n >= 0 && m >= 0, BQ_EINVAL, "three_interpolate_grad: bad extents");
  if (B == 0 || C == 0 || n == 0) return BQ_OK;
  BQ_REQUIRE(grad_out && idx && weight && grad_points, BQ_EINVAL, "three_interpolate_grad: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "three_interpolate_grad: B > 65535");
  const int xb = cdiv(n, 256);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(xb, cgrid(C, xb, B), B), dim3(256), 0, (hipStream_t)stream,
                     grad_out, idx, weight, grad_points, C, n, m);
  return check_launch("three_interpolate_grad");
}

// ---- diagnostics -------------------------------------------------------------------------------
// Shader clock seen by a resident wave: d(s_memtime) / d(s_memrealtime) * 100 MHz (MI355X_MICROARCH.md,
// 'DVFS give-back' item 6).  out[0] = MHz measured by block 0 after `iters` dependent FMAs per lane.
__global__ void debug_clock_kernel(float *out, int iters) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float v = (float)threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1.0000001f + 0.5f;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (float)((double)(c1 - c0) / (double)(r1 - r0) * 100.0);
    out[1] = v;
    out[2] = (float)(c1 - c0);
  }
}

extern "C" __attribute__((visibility("default"))) int bq_debug_clock_mhz(float *out, int blocks, int iters,
                                                                         void *stream) {
  hipLaunchKernelGGL(debug_clock_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return bq::check_launch("debug_clock");
}


// Point-major grouping (see group_concat_pm_kernel).  feats: f32 rows of C contiguous elements, batch stride f_bs
// and row stride f_rs in elements (so a (B,N,3+C) interleaved cloud can be read in place); out: (B,M,S,3+C) f32 or
// bf16.
extern "C" __attribute__((visibility("default"))) int bq_group_concat_pm(
    const float *xyz, const float *new_xyz, const float *feats, long f_bs, long f_rs, const int32_t *idx, void *out,
    int out_bf16, int B, int C, int N, int M, int S, float radius, int normalize, int ld, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_concat_pm: bad extents");
  BQ_REQUIRE(ld >= C + 3 && ld - (C + 3) < 64, BQ_EINVAL, "group_concat_pm: row stride %d for %d channels", ld, C + 3);
  if (B == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(xyz && new_xyz && idx && out && (feats || C == 0), BQ_EINVAL, "group_concat_pm: null pointer");
  const long total = (long)B * M * S;
  const int blocks = (int)(total / 4 < 8192 ? (total + 3) / 4 : 8192);
  if (out_bf16 && C > 0 && C % 4 == 0 && C <= 256 && ld <= 264 && ld % 8 == 0 && S % GC_RW == 0 && ((uintptr_t)out % 16 == 0) &&
      ((uintptr_t)feats % 4 == 0)) {
    const long strips = total / GC_RW;
    const int vblocks = (int)(strips / 4 < 8192 ? (strips + 3) / 4 : 8192);
    hipLaunchKernelGGL(group_concat_pm_vec_kernel, dim3(vblocks), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, feats, f_bs,
                       f_rs, idx, (__bf16 *)out, C, N, M, S, radius, normalize, total, ld);
  } else if (out_bf16)
    hipLaunchKernelGGL(group_concat_pm_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz,
                       feats, f_bs, f_rs, idx, (__bf16 *)out, C, N, M, S, radius, normalize, total, ld);
  else
    hipLaunchKernelGGL(group_concat_pm_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz,
                       feats, f_bs, f_rs, idx, (float *)out, C, N, M, S, radius, normalize, total, ld);
  return check_launch("group_concat_pm");
}

// grad_out: (B,M,S,3+C) f32 or bf16; grad_feats (B,N,C) f32, grad_xyz (B,N,3), grad_new_xyz (B,M,3): zero_init, each
// optional (NULL).
extern "C" __attribute__((visibility("default"))) int bq_group_concat_pm_grad(
    const void *grad_out, int in_bf16, const int32_t *idx, float *grad_feats, float *grad_xyz, float *grad_new_xyz,
    int B, int C, int N, int M, int S, float radius, int normalize, int ld, void *stream) {
  BQ_REQUIRE(B >= 0 && C >= 0 && N >= 0 && M >= 0 && S >= 0, BQ_EINVAL, "group_concat_pm_grad: bad extents");
  BQ_REQUIRE(ld >= C + 3, BQ_EINVAL, "group_concat_pm_grad: row stride %d for %d channels", ld, C + 3);
  if (B == 0 || M == 0 || S == 0) return BQ_OK;
  BQ_REQUIRE(grad_out && idx, BQ_EINVAL, "group_concat_pm_grad: null pointer");
  const long total = (long)B * M * S;
  const int blocks = (int)(total / 4 < 8192 ? (total + 3) / 4 : 8192);
  if (in_bf16)
    hipLaunchKernelGGL(group_concat_pm_grad_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)grad_out, idx, grad_feats, grad_xyz, grad_new_xyz, C, N, M, S, radius,
                       normalize, total, ld);
  else
    hipLaunchKernelGGL(group_concat_pm_grad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float *)grad_out, idx, grad_feats, grad_xyz, grad_new_xyz, C, N, M, S, radius,
                       normalize, total, ld);
  return check_launch("group_concat_pm_grad");
}
