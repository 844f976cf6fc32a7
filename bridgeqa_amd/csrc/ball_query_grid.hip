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
// The launches:
//   grid_build_kernel    one workgroup per scene: bounding box, grid (cell edge >= 2 radius, <= GRID_MAXC cells), counting
//                        sort of the points by cell into (x, y, z, index) records + the cell start table (global scratch)
//                        -- round 5's build, still the one for N < 4096; from there on grid_box_kernel, grid_cellid_kernel
//                        and grid_chunk_kernel (below) build the same tables with 16 workgroups per scene;
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

// ---- the same build with SEVERAL workgroups per scene (round 6) ---------------------------------------------
// One workgroup per scene walks 40 000 points three times (a CU draws ~50 GB/s: ~10 us a pass) and scatters 40 000 16-byte
// records through ONE CU's address unit: 50 us, twice the query it serves.  Three short launches instead:
//   grid_box_kernel     GRID_NWG workgroups per scene: partial bounding boxes of slices of the points;
//   grid_cellid_kernel  elementwise over the points: every workgroup derives the SAME grid from the partial boxes (min / max are
//                       exact and order-independent) and stores the cell of each point as 16 bits;
//   grid_chunk_kernel   GRID_NWG workgroups per scene, each owning a CHUNK OF CELLS (ncell / GRID_NWG consecutive cells, hence
//                       a contiguous range of the records) and needing nothing from the others: it reads the scene's cell ids
//                       (2 bytes a point, not 12), histograms the points of its cells in LDS and counts the points of EARLIER
//                       cells -- where its range of the records begins, no scan across workgroups -- scans its <= 512
//                       counters into the start table and its cursors, then reads the ids again, fetches the coordinates of
//                       its own points and writes their records.  Its workgroups of one scene sit on ONE XCD (workgroup L runs
//                       on XCD L % 8): the ids and coordinates they share are fetched into one L2.
// What was tried first (all index-exact, tools/time_ball_query.py, build alone at SA1's shape):
//   slices of the points per workgroup, device-scope atomics on one table per scene           110 us (they execute at the memory
//                                                                                                 side on this multi-XCD part)
//   slices, LDS histograms, a (16 x cells) table, four launches (box / count / scan / scatter)  28 us (5.0 + 5.7 + 5.1 + 12.4; the
//            scan was 50 us element by element and 23.5 us batched while ONE workgroup per scene pulled the table through one
//            CU; the scatter's 16-byte records land in lines shared with other XCDs' slices: partial-line write-backs)
//   cell chunks in ONE launch, every workgroup reading all the points three times                36 us (3 x 480 KB per CU)
// The order of the records inside a cell is whatever the LDS atomics give -- the query kernel ranks the hits of a ball by
// point index, so its output does not depend on it.
constexpr int GRID_NWG = 16;
constexpr int GRID_CHUNK = GRID_MAXC / GRID_NWG;
constexpr int GRID_MULTI_MAXN = 1 << 19;   // (a point index and a cell of the chunk share a 32-bit list entry)

__device__ __forceinline__ GridHeader grid_header(const float *box, int N, float radius) {
  GridHeader g;
  g.lox = box[0]; g.loy = box[1]; g.loz = box[2];
  const float ex = fmaxf(box[3] - g.lox, 1e-6f), ey = fmaxf(box[4] - g.loy, 1e-6f), ez = fmaxf(box[5] - g.loz, 1e-6f);
  const float ce = fmaxf(2.0f * radius, 1e-6f);     // (as in grid_build_kernel: any grid is exact)
  g.gx = min(max((int)(ex / ce), 1), 1024);
  g.gy = min(max((int)(ey / ce), 1), 1024);
  g.gz = min(max((int)(ez / ce), 1), 1024);
  while ((long)g.gx * g.gy * g.gz > GRID_MAXC) {
    if (g.gx >= g.gy && g.gx >= g.gz) g.gx = (g.gx + 1) / 2; else if (g.gy >= g.gz) g.gy = (g.gy + 1) / 2; else g.gz = (g.gz + 1) / 2;
  }
  g.invx = (float)g.gx / ex; g.invy = (float)g.gy / ey; g.invz = (float)g.gz / ez;
  g.n = N;
  return g;
}
__device__ __forceinline__ int grid_cell(const GridHeader &g, float x, float y, float z) {
  return (cell1(z, g.loz, g.invz, g.gz) * g.gy + cell1(y, g.loy, g.invy, g.gy)) * g.gx + cell1(x, g.lox, g.invx, g.gx);
}
// per-scene scratch behind the records: GRID_NWG partial boxes (6 floats each), then the cell ids (16 bits a point, whole uint4s)
__host__ __device__ __forceinline__ size_t grid_scratch_floats(int N) { return GRID_NWG * 6 + 4 * (((size_t)N + 7) / 8); }
__device__ __forceinline__ float *grid_parts(float *base, int N) { return base + GRID_HDR + GRID_MAXC + 16 + 4 * (size_t)N; }
__device__ __forceinline__ unsigned short *grid_ids(float *base, int N) {
  return reinterpret_cast<unsigned short *>(grid_parts(base, N) + GRID_NWG * 6);
}

__global__ __launch_bounds__(256) void grid_box_kernel(const float *__restrict__ xyz, float *__restrict__ ws, int N, size_t ws_stride) {
  __shared__ float s_red[4][6];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, w = blockIdx.x;
  const float *P = xyz + (size_t)blockIdx.y * N * 3;
  float *base = ws + (size_t)blockIdx.y * ws_stride;
  const float BIG = 3.0e38f;
  float lx = BIG, ly = BIG, lz = BIG, hx = -BIG, hy = -BIG, hz = -BIG;
  const int per = (N + GRID_NWG - 1) / GRID_NWG, k1 = min(N, (w + 1) * per);
  constexpr int U = 8;   // (loads of a batch in flight together: one memory round trip per 8 points, not per point)
  for (int k0 = w * per + t; k0 < k1; k0 += 256 * U) {
    f32x3 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = *reinterpret_cast<const f32x3 *>(P + (size_t)min(k0 + u * 256, k1 - 1) * 3);
#pragma unroll
    for (int u = 0; u < U; ++u) {   // (a repeated point changes no minimum / maximum)
      lx = fminf(lx, p[u][0]); ly = fminf(ly, p[u][1]); lz = fminf(lz, p[u][2]);
      hx = fmaxf(hx, p[u][0]); hy = fmaxf(hy, p[u][1]); hz = fmaxf(hz, p[u][2]);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    lx = fminf(lx, __shfl_xor(lx, off)); ly = fminf(ly, __shfl_xor(ly, off)); lz = fminf(lz, __shfl_xor(lz, off));
    hx = fmaxf(hx, __shfl_xor(hx, off)); hy = fmaxf(hy, __shfl_xor(hy, off)); hz = fmaxf(hz, __shfl_xor(hz, off));
  }
  if (lane == 0) { s_red[wid][0] = lx; s_red[wid][1] = ly; s_red[wid][2] = lz; s_red[wid][3] = hx; s_red[wid][4] = hy; s_red[wid][5] = hz; }
  __syncthreads();
  if (t < 6) {   // (a slice past the last point keeps +-BIG: neutral)
    float v = s_red[0][t];
    for (int i = 1; i < 4; ++i) v = t < 3 ? fminf(v, s_red[i][t]) : fmaxf(v, s_red[i][t]);
    grid_parts(base, N)[w * 6 + t] = v;
  }
}

constexpr int GRID_IDS_T = 256, GRID_IDS_U = 4;   // points per workgroup of grid_cellid_kernel: T x U
__global__ __launch_bounds__(GRID_IDS_T) void grid_cellid_kernel(const float *__restrict__ xyz, float *__restrict__ ws, int N, float radius,
                                                                 size_t ws_stride) {
  __shared__ float s_box[6];
  constexpr int T = GRID_IDS_T, U = GRID_IDS_U;
  const int t = threadIdx.x;
  const float *P = xyz + (size_t)blockIdx.y * N * 3;
  float *base = ws + (size_t)blockIdx.y * ws_stride;
  const int k0 = blockIdx.x * T * U + t;
  f32x3 p[U];
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = *reinterpret_cast<const f32x3 *>(P + (size_t)min(k0 + u * T, N - 1) * 3);
  if (t < 6) {
    const float *parts = grid_parts(base, N);
    float v = parts[t];
    for (int w = 1; w < GRID_NWG; ++w) v = t < 3 ? fminf(v, parts[w * 6 + t]) : fmaxf(v, parts[w * 6 + t]);
    s_box[t] = v;
  }
  __syncthreads();
  float box[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) box[c] = s_box[c];
  const GridHeader g = grid_header(box, N, radius);
  if (blockIdx.x == 0 && t == 0) {
    base[0] = g.lox; base[1] = g.loy; base[2] = g.loz; base[3] = g.invx; base[4] = g.invy; base[5] = g.invz;
    reinterpret_cast<int *>(base)[6] = g.gx; reinterpret_cast<int *>(base)[7] = g.gy; reinterpret_cast<int *>(base)[8] = g.gz;
    reinterpret_cast<int *>(base)[9] = N;
    reinterpret_cast<int *>(base + GRID_HDR)[g.gx * g.gy * g.gz] = N;   // the end marker of the start table
  }
  unsigned short *ids = grid_ids(base, N);
  const int npad = ((N + 7) / 8) * 8;   // (<= the points this launch covers: a multiple of T x U)
#pragma unroll
  for (int u = 0; u < U; ++u) {   // the ids past N in the last group of 8 name no cell
    const int k = k0 + u * T;
    if (k < npad) ids[k] = k < N ? (unsigned short)grid_cell(g, p[u][0], p[u][1], p[u][2]) : (unsigned short)0xffffu;
  }
}

__global__ __launch_bounds__(1024) void grid_chunk_kernel(const float *__restrict__ xyz, float *__restrict__ ws, int B, int N,
                                                          size_t ws_stride) {
  constexpr int T = 1024, U = 5, H = 8;
  __shared__ int s_cnt[GRID_CHUNK];
  __shared__ int s_below[16], s_wsum[16];
  __shared__ unsigned s_list[H * T];   // per thread: its first H points of the chunk (index | cell << 19), fetched together
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  // workgroup L -> XCD L % 8; the GRID_NWG workgroups of a scene share an XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int scene = (slot / GRID_NWG) * 8 + xcd, j = slot % GRID_NWG;
  if (scene >= B) return;
  const float *P = xyz + (size_t)scene * N * 3;
  float *base = ws + (size_t)scene * ws_stride;
  int *starts = reinterpret_cast<int *>(base + GRID_HDR);
  float4 *rec = reinterpret_cast<float4 *>(base + GRID_HDR + GRID_MAXC + 16);
  const uint4 *ids = reinterpret_cast<const uint4 *>(grid_ids(base, N));
  const int nvec = (N + 7) / 8;   // (the ids past N in the last uint4 are 0xffff: in no chunk, before none)
  const int ncell = reinterpret_cast<const int *>(base)[6] * reinterpret_cast<const int *>(base)[7] * reinterpret_cast<const int *>(base)[8];
  const int chunk = (ncell + GRID_NWG - 1) / GRID_NWG;          // <= GRID_CHUNK
  const int c0 = j * chunk, nc = min(ncell, c0 + chunk) - c0;    // this workgroup's cells: [c0, c0 + nc)
  if (nc <= 0) return;   // (uniform)
  if (t < GRID_CHUNK) s_cnt[t] = 0;
  __syncthreads();
  auto id_of = [](const uint4 &q, int e) { const unsigned w = e < 2 ? q.x : e < 4 ? q.y : e < 6 ? q.z : q.w; return (int)((e & 1) ? w >> 16 : w & 0xffffu); };
  int below = 0, mine = 0;
  for (int v0 = t; v0 < nvec; v0 += T * U) {
    uint4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) q[u] = v0 + u * T < nvec ? ids[v0 + u * T] : make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int rel = id_of(q[u], e) - c0;
        below += rel < 0;
        if ((unsigned)rel < (unsigned)nc) {
          atomicAdd(&s_cnt[rel], 1);
          if (mine < H) s_list[mine * T + t] = (unsigned)((v0 + u * T) * 8 + e) | ((unsigned)rel << 19);
          ++mine;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) below += __shfl_xor(below, off);
  if (lane == 0) s_below[wid] = below;
  const int crowded = __syncthreads_or(mine > H);   // (a dense chunk: some thread met more points than it could list)
  {   // exclusive scan of the chunk's counters, one cell per thread
    static_assert(GRID_CHUNK <= T, "one thread per cell of the chunk");
    const int v = t < nc ? s_cnt[t] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wsum[wid] = incl;
    __syncthreads();
    int start = incl - v;
    for (int w = 0; w < 16; ++w) start += s_below[w] + (w < wid ? s_wsum[w] : 0);
    if (t < nc) { starts[c0 + t] = start; s_cnt[t] = start; }
  }
  __syncthreads();
  auto place = [&](int k, int rel, const f32x3 &p) {
    const int pos = atomicAdd(&s_cnt[rel], 1);
    rec[pos] = make_float4(p[0], p[1], p[2], __int_as_float(k));
  };
  {   // the listed points: their coordinates are fetched together (one memory round trip, not one per point)
    f32x3 p[H];
    unsigned en[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      en[i] = i < mine ? s_list[i * T + t] : 0u;   // (own entries only)
      p[i] = *reinterpret_cast<const f32x3 *>(P + (size_t)(en[i] & 0x7ffffu) * 3);
    }
#pragma unroll
    for (int i = 0; i < H; ++i)
      if (i < mine) place((int)(en[i] & 0x7ffffu), (int)(en[i] >> 19), p[i]);
  }
  if (crowded) {   // the points past a thread's list: the ids once more, one point at a time
    int seen = 0;
    for (int v0 = t; v0 < nvec; v0 += T * U) {
      uint4 q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) q[u] = v0 + u * T < nvec ? ids[v0 + u * T] : make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = (v0 + u * T) * 8 + e;
          const int rel = id_of(q[u], e) - c0;
          if ((unsigned)rel < (unsigned)nc) {
            if (seen >= H) place(k, rel, *reinterpret_cast<const f32x3 *>(P + (size_t)k * 3));
            ++seen;
          }
        }
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

// measurement switch (tools/time_ball_query.py): 0 = the one-workgroup-per-scene build of round 5
static int g_grid_build_multi = 1;
extern "C" int bq_ball_query_grid_build_mode(int multi) {
  const int prev = g_grid_build_multi;
  g_grid_build_multi = multi;
  return prev;
}

extern "C" size_t bq_ball_query_grid_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  // floats: header, start table, records, and (round 6) the partial boxes + the cell ids of the multi-workgroup build
  const size_t per = (size_t)bq::GRID_HDR + bq::GRID_MAXC + 16 + 4 * (size_t)N + bq::grid_scratch_floats(N);
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
  const size_t stride = (size_t)GRID_HDR + GRID_MAXC + 16 + 4 * (size_t)N + grid_scratch_floats(N);
  const float radius2 = radius * radius;   // ball_query_gpu.cu:22, rounded to fp32 on the host
  float *ws = reinterpret_cast<float *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  if (g_grid_build_multi && N >= 4096 && N < GRID_MULTI_MAXN) {
    hipLaunchKernelGGL(grid_box_kernel, dim3(GRID_NWG, B), dim3(256), 0, st, xyz, ws, N, stride);
    const int per_wg = GRID_IDS_T * GRID_IDS_U;
    hipLaunchKernelGGL(grid_cellid_kernel, dim3((N + per_wg - 1) / per_wg, B), dim3(GRID_IDS_T), 0, st, xyz, ws, N, radius, stride);
    const int groups = (B + 7) / 8;   // 8 scenes (one per XCD) x GRID_NWG workgroups each
    hipLaunchKernelGGL(grid_chunk_kernel, dim3(groups * GRID_NWG * 8), dim3(1024), 0, st, xyz, ws, B, N, stride);
  } else {
    hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(1024), 0, st, xyz, ws, N, radius, stride);
  }
  int gx = (M + 3) / 4;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(ball_query_grid_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, ws, idx, N, M, radius,
                     radius2, nsample, stride);
  return check_launch("ball_query_grid");
}
