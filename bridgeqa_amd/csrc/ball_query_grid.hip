// ball_query for LARGE scenes through a uniform grid -- index-exact with the reference's exhaustive scan
// (lib/pointnet2/_ext_src/src/ball_query_gpu.cu:9-44: for every centre the first `nsample` points IN INDEX ORDER with
// d2 < radius^2, the remaining slots filled with the first hit, all zeros when the ball is empty).
//
// Why: the wave-per-centre-pair kernel of pn2_ops.hip scans all N points for every centre -- 1.31 G
// point-centre pairs at SA1 (B = 16, N = 40000, M = 2048): 0.53 ms alone, at the vector-instruction bound of the scan, and
// 2.1 ms on the gentle grid it runs on beside the fusion chain, where it still costs the step 0.9 ms of interference
// (tools/geo_probe.py).  A ball of SA1's radius holds a few dozen points; a grid with cells of twice the radius leaves
// ~100 candidates per centre: ~400x fewer distance tests.
//
// Two launches:
//   grid_build_kernel    one workgroup per scene: bounding box, grid (cell edge >= 2 radius, <= GRID_MAXC cells), counting
//                        sort of the points by cell into (x, y, z, index) records + the cell start table (global scratch);
//   ball_query_grid_kernel  one wave per centre: the cells that meet [c - s, c + s]^3 (s a hair above the radius; the cell
//                        of a coordinate is a monotone function of it, so every point with d2 < r^2 lies in a visited cell),
//                        runs of consecutive cells (same z, y) as one coalesced sweep; every candidate gets the CANONICAL
//                        distance test (bq_common.h sqdist: the very comparison of the exhaustive kernel and of the oracle),
//                        hits are collected as point indices in LDS, ranked (rank = number of smaller indices) and the
//                        first nsample written in ascending index order -- what the serial scan would have produced.
//                        A ball with more than GRID_CAP hits (dense real scans) falls back to the exhaustive scan of that
//                        centre, which stops after nsample hits -- early, since such a ball is dense.
#include "bq_common.h"

namespace bq {

constexpr int GRID_MAXC = 8192;    // cells per scene (LDS histogram of the build kernel)
constexpr int GRID_CAP = 256;      // in-ball indices a wave can rank in LDS
constexpr int GRID_HDR = 16;       // floats: lo x y z, inv x y z, gx gy gz (as ints), N

typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));

struct GridHeader { float lox, loy, loz, invx, invy, invz; int gx, gy, gz, n; };

__device__ __forceinline__ int cell1(float x, float lo, float inv, int g) {
  // monotone non-decreasing in x (subtraction, product, clamp and truncation all are)
  const int c = (int)fminf(fmaxf((x - lo) * inv, -1.0f), 2048.0f);
  return min(max(c, 0), g - 1);
}

__global__ __launch_bounds__(1024) void grid_build_kernel(const float *__restrict__ xyz, float *__restrict__ ws, int N, float radius,
                                                          size_t ws_stride) {
  __shared__ int s_cnt[GRID_MAXC + 2];
  __shared__ float s_red[16][6];
  __shared__ int s_wsum[16];
  __shared__ float s_box[6];
  constexpr int T = 1024;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const float *P = xyz + (size_t)blockIdx.x * N * 3;
  float *base = ws + (size_t)blockIdx.x * ws_stride;
  int *starts = reinterpret_cast<int *>(base + GRID_HDR);
  float4 *rec = reinterpret_cast<float4 *>(base + GRID_HDR + GRID_MAXC + 16);
  const float BIG = 3.0e38f;
  float lx = BIG, ly = BIG, lz = BIG, hx = -BIG, hy = -BIG, hz = -BIG;
  // (every pass over the points takes them U at a time: the loads of a batch are in flight together -- one point per
  // iteration was a memory round trip per point and thread, 39 in a row per pass at N = 40 000: 65 us for the launch)
  constexpr int U = 8;
  for (int k0 = t; k0 < N; k0 += T * U) {
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * T, N - 1);   // (a repeated last point changes no minimum / maximum)
      const f32x3 p3 = *reinterpret_cast<const f32x3 *>(P + k * 3);   // one 12-byte load per point
      px[u] = p3[0]; py[u] = p3[1]; pz[u] = p3[2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      lx = fminf(lx, px[u]); ly = fminf(ly, py[u]); lz = fminf(lz, pz[u]);
      hx = fmaxf(hx, px[u]); hy = fmaxf(hy, py[u]); hz = fmaxf(hz, pz[u]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    lx = fminf(lx, __shfl_xor(lx, off)); ly = fminf(ly, __shfl_xor(ly, off)); lz = fminf(lz, __shfl_xor(lz, off));
    hx = fmaxf(hx, __shfl_xor(hx, off)); hy = fmaxf(hy, __shfl_xor(hy, off)); hz = fmaxf(hz, __shfl_xor(hz, off));
  }
  if (lane == 0) { s_red[wid][0] = lx; s_red[wid][1] = ly; s_red[wid][2] = lz; s_red[wid][3] = hx; s_red[wid][4] = hy; s_red[wid][5] = hz; }
  for (int c = t; c < GRID_MAXC + 2; c += T) s_cnt[c] = 0;
  __syncthreads();
  if (t < 6) {
    float v = s_red[0][t];
    for (int w = 1; w < 16; ++w) v = t < 3 ? fminf(v, s_red[w][t]) : fmaxf(v, s_red[w][t]);
    s_box[t] = v;
  }
  __syncthreads();
  GridHeader g;
  {
    g.lox = s_box[0]; g.loy = s_box[1]; g.loz = s_box[2];
    const float ex = fmaxf(s_box[3] - g.lox, 1e-6f), ey = fmaxf(s_box[4] - g.loy, 1e-6f), ez = fmaxf(s_box[5] - g.loz, 1e-6f);
    const float ce = fmaxf(2.0f * radius, 1e-6f);     // (efficiency only: any grid is exact, see the visit range below)
    g.gx = min(max((int)(ex / ce), 1), 1024);
    g.gy = min(max((int)(ey / ce), 1), 1024);
    g.gz = min(max((int)(ez / ce), 1), 1024);
    while ((long)g.gx * g.gy * g.gz > GRID_MAXC) {   // uniform: every thread computes the same values
      if (g.gx >= g.gy && g.gx >= g.gz) g.gx = (g.gx + 1) / 2; else if (g.gy >= g.gz) g.gy = (g.gy + 1) / 2; else g.gz = (g.gz + 1) / 2;
    }
    g.invx = (float)g.gx / ex; g.invy = (float)g.gy / ey; g.invz = (float)g.gz / ez;
    g.n = N;
  }
  const int ncell = g.gx * g.gy * g.gz;
  auto cell_of = [&](float x, float y, float z) {
    return (cell1(z, g.loz, g.invz, g.gz) * g.gy + cell1(y, g.loy, g.invy, g.gy)) * g.gx + cell1(x, g.lox, g.invx, g.gx);
  };
  for (int k0 = t; k0 < N; k0 += T * U) {
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * T, N - 1);
      const f32x3 p3 = *reinterpret_cast<const f32x3 *>(P + k * 3);   // one 12-byte load per point
      px[u] = p3[0]; py[u] = p3[1]; pz[u] = p3[2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (k0 + u * T < N) atomicAdd(&s_cnt[cell_of(px[u], py[u], pz[u])], 1);
  }
  __syncthreads();
  {   // exclusive scan of the counters by the whole workgroup: thread t owns cells [8 t, 8 t + 8) (GRID_MAXC = 8 x 1024), wave
      // scans of the per-thread totals, one 16-entry scan of the wave totals (round 5 had ONE wave walk the table 64 cells at
      // a time: 128 dependent six-step shuffle scans, ~25 of the launch's 65 us)
    static_assert(GRID_MAXC == 8 * T, "one thread per 8 cells");
    int v[8], tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = (8 * t + i < ncell) ? s_cnt[8 * t + i] : 0;
      tot += v[i];
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    __syncthreads();                       // (every thread has read its cells; s_wsum aliases nothing)
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_wsum[w];
    int base = woff + incl - tot;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (8 * t + i <= ncell) s_cnt[8 * t + i] = base;
      base += v[i];
    }
    if (t == T - 1 && ncell == GRID_MAXC) s_cnt[GRID_MAXC] = base;   // (the end marker of a full table lies past the last thread's cells)
  }
  __syncthreads();
  for (int c = t; c <= ncell; c += T) starts[c] = s_cnt[c];
  if (t == 0) {
    base[0] = g.lox; base[1] = g.loy; base[2] = g.loz; base[3] = g.invx; base[4] = g.invy; base[5] = g.invz;
    reinterpret_cast<int *>(base)[6] = g.gx; reinterpret_cast<int *>(base)[7] = g.gy; reinterpret_cast<int *>(base)[8] = g.gz;
    reinterpret_cast<int *>(base)[9] = N;
  }
  __syncthreads();   // (the table is copied out before the scatter advances the counters)
  // (order inside a cell is whatever the atomics give: the query kernel ranks the hits of a ball by point index)
  for (int k0 = t; k0 < N; k0 += T * U) {
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * T, N - 1);
      const f32x3 p3 = *reinterpret_cast<const f32x3 *>(P + k * 3);   // one 12-byte load per point
      px[u] = p3[0]; py[u] = p3[1]; pz[u] = p3[2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * T;
      if (k < N) {
        const int pos = atomicAdd(&s_cnt[cell_of(px[u], py[u], pz[u])], 1);
        rec[pos] = make_float4(px[u], py[u], pz[u], __int_as_float(k));
      }
    }
  }
}

__global__ __launch_bounds__(256) void ball_query_grid_kernel(const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                              const float *__restrict__ ws, int32_t *__restrict__ idx, int N, int M,
                                                              float radius, float radius2, int S, size_t ws_stride) {
  __shared__ int s_list[4][GRID_CAP];
  const int b = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float *base = ws + (size_t)b * ws_stride;
  const int *starts = reinterpret_cast<const int *>(base + GRID_HDR);
  const float4 *rec = reinterpret_cast<const float4 *>(base + GRID_HDR + GRID_MAXC + 16);
  const float lox = base[0], loy = base[1], loz = base[2], invx = base[3], invy = base[4], invz = base[5];
  const int gx = reinterpret_cast<const int *>(base)[6], gy = reinterpret_cast<const int *>(base)[7], gz = reinterpret_cast<const int *>(base)[8];
  // a hair above the radius: d2 < fl(r r) implies |dx| <= sqrt(d2) < r (1 + 2^-23) in exact arithmetic, and fl(x - c) is within
  // half an ulp of the coordinate difference; 1e-3 relative + 1e-6 absolute covers both by orders of magnitude
  const float s = radius * 1.001f + 1e-6f;
  int *lst = s_list[w];
  for (int j = blockIdx.x * 4 + w; j < M; j += gridDim.x * 4) {
    const float *q = new_xyz + ((size_t)b * M + j) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int32_t *o = idx + ((size_t)b * M + j) * S;
    const int cx0 = cell1(qx - s, lox, invx, gx), cx1 = cell1(qx + s, lox, invx, gx);
    const int cy0 = cell1(qy - s, loy, invy, gy), cy1 = cell1(qy + s, loy, invy, gy);
    const int cz0 = cell1(qz - s, loz, invz, gz), cz1 = cell1(qz + s, loz, invz, gz);
    int cnt = 0;
    for (int cz = cz0; cz <= cz1 && cnt <= GRID_CAP; ++cz)
      for (int cy = cy0; cy <= cy1 && cnt <= GRID_CAP; ++cy) {
        const int c0 = (cz * gy + cy) * gx;
        const int p0 = starts[c0 + cx0], p1 = starts[c0 + cx1 + 1];   // cells cx0 .. cx1 of this row are contiguous
        for (int pb = p0; pb < p1 && cnt <= GRID_CAP; pb += 64) {   // (wave-uniform trip count)
          const int p = pb + lane;
          const bool in = p < p1;
          const float4 r = rec[in ? p : p1 - 1];
          const bool hit = in && sqdist(qx, qy, qz, r.x, r.y, r.z) < radius2;
          const unsigned long long m = __ballot(hit);
          if (m) {
            const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (hit && pos < GRID_CAP) lst[pos] = __float_as_int(r.w);
            cnt += __builtin_popcountll(m);
          }
        }
      }
    if (cnt > GRID_CAP) {
      // a dense ball: the exhaustive scan in index order, which ends after S hits (ball_query_wave_kernel's loop, one centre)
      const float *P = xyz + (size_t)b * N * 3;
      int c2 = 0, first = 0;
      for (int k0 = 0; k0 < N && c2 < S; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < N;
        const int kk = in ? k : N - 1;
        const bool hit = in && sqdist(qx, qy, qz, P[kk * 3], P[kk * 3 + 1], P[kk * 3 + 2]) < radius2;
        const unsigned long long m = __ballot(hit);
        if (m) {
          if (c2 == 0) first = k0 + __builtin_ctzll(m);
          const int pos = c2 + __builtin_popcountll(m & ((1ull << lane) - 1ull));
          if (hit && pos < S) o[pos] = k;
          c2 += __builtin_popcountll(m);
        }
      }
      if (c2 > S) c2 = S;
      for (int e = c2 + lane; e < S; e += 64) o[e] = first;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS stores are visible to its lanes
    // rank of every hit among the hits = its slot in index order
    int smallest = 0x7fffffff;
    for (int e0 = 0; e0 < cnt; e0 += 64) {
      const int e = e0 + lane;
      const int mine = e < cnt ? lst[e] : 0x7fffffff;
      int rank = 0;
      for (int f = 0; f < cnt; ++f) rank += lst[f] < mine ? 1 : 0;   // (a broadcast LDS read per step)
      if (e < cnt && rank < S) o[rank] = mine;
      smallest = min(smallest, mine);
    }
    for (int off = 32; off > 0; off >>= 1) smallest = min(smallest, __shfl_xor(smallest, off));
    const int first = cnt ? smallest : 0;   // the serial scan's first hit = the smallest index in the ball
    for (int e = min(cnt, S) + lane; e < S; e += 64) o[e] = first;
    __builtin_amdgcn_wave_barrier();        // (the list is rewritten by the next centre)
  }
}

}  // namespace bq

extern "C" size_t bq_ball_query_grid_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  const size_t per = (size_t)bq::GRID_HDR + bq::GRID_MAXC + 16 + 4 * (size_t)N;   // floats
  return (size_t)B * per * sizeof(float);
}

extern "C" int bq_ball_query_grid(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                                  int nsample, void *workspace, size_t workspace_bytes, void *stream) {
  using namespace bq;
  BQ_REQUIRE(B >= 0 && N >= 0 && M >= 0 && nsample >= 0, BQ_EINVAL, "ball_query_grid: bad extents");
  if (B == 0 || M == 0 || nsample == 0) return 0;
  BQ_REQUIRE(N > 0, BQ_EINVAL, "ball_query_grid: empty scene (use bq_ball_query)");
  BQ_REQUIRE(new_xyz && xyz && idx, BQ_EINVAL, "ball_query_grid: null pointer");
  BQ_REQUIRE(B <= 65535, BQ_ELIMIT, "ball_query_grid: B=%d > 65535", B);
  BQ_REQUIRE(radius > 0.0f && radius < 1e18f, BQ_EINVAL, "ball_query_grid: radius must be positive and finite");
  BQ_REQUIRE(workspace && workspace_bytes >= bq_ball_query_grid_workspace_bytes(B, N), BQ_EINVAL,
             "ball_query_grid: workspace of %zu bytes required", bq_ball_query_grid_workspace_bytes(B, N));
  const size_t stride = (size_t)GRID_HDR + GRID_MAXC + 16 + 4 * (size_t)N;
  const float radius2 = radius * radius;   // ball_query_gpu.cu:22, rounded to fp32 on the host
  float *ws = reinterpret_cast<float *>(workspace);
  hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, xyz, ws, N, radius, stride);
  int gx = (M + 3) / 4;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(ball_query_grid_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, ws, idx, N, M, radius,
                     radius2, nsample, stride);
  return check_launch("ball_query_grid");
}
