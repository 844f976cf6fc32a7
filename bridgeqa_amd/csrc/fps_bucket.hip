// Bucketed, pruned -- and still index-exact -- furthest point sampling for large scenes on gfx950.
//
// Why: the reference (sampling_gpu.cu:69-173) touches every point in every round: 20*N*(m-1)
// bytes per scene, 26 GB for SA1 at B=16.  But a round only CHANGES the running minimum distance
// of points that are closer to the new sample than to every earlier one -- a shrinking
// neighbourhood.  This kernel keeps the scene sorted into spatially compact ROWS of 64 points
// (one wave-wide register row each) with a bounding box per row and skips every row whose box is
// provably too far away to be changed.
//
// Exactness.  For a row with box [lo,hi] and new sample p let g_a = max(fl(lo_a-p_a), fl(p_a-hi_a), 0)
// and LB = fl(fl(fl(g_x*g_x)+fl(g_y*g_y))+fl(g_z*g_z)), evaluated with the same separately-rounded
// operations as the distance itself.  Rounding is monotone, so every point q of the row has
// |fl(q_a-p_a)| >= g_a and therefore d(q,p) >= LB *in floating point*, no epsilon.  If
// LB >= max_row(min_dist) then min(min_dist[q], d) == min_dist[q] for the whole row: skipping it
// leaves every value -- and the row's cached arg-max -- exactly what the reference would hold.
// The arg-max itself is taken on the total order (d desc, bitrev(k mod bs) asc, k asc) (see
// fps.hip), so neither the sorted layout nor the reduction shape can change the result.
//
// Layout: one workgroup (NW waves) per scene.  Phase 1 sorts the scene by a snake-ordered uniform
// grid cell (LDS histogram + scan + scatter) into a structure-of-arrays scratch in global memory
// (L2 resident: 16 B/point).  Row r belongs to wave r % NW; its record (box, row max, tie key and
// coordinates of its best point) lives in the REGISTERS of lane (r / NW) % 64 of that wave.  The
// running minima live in LDS when the scene fits (N <= ~40 k), else next to the sorted points.
// Per round: every lane tests its rows (registers only) -> ballot -> the wave walks its active
// rows (3+1 coalesced 256-B loads each) -> wave-level then workgroup-level arg-max through LDS
// with ONE barrier.
#include "bq_common.h"

namespace bq {

__device__ __forceinline__ unsigned tie_key2(int k, int log2bs) {
  const unsigned cls = log2bs ? (__brev((unsigned)k & ((1u << log2bs) - 1u)) >> (32 - log2bs)) : 0u;
  return (cls << 22) | (unsigned)k;
}

__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_f32<0x111, 0xF>(v));
  v = fminf(v, dpp_f32<0x112, 0xF>(v));
  v = fminf(v, dpp_f32<0x114, 0xF>(v));
  v = fminf(v, dpp_f32<0x118, 0xF>(v));
  v = fminf(v, dpp_f32<0x142, 0xA>(v));
  v = fminf(v, dpp_f32<0x143, 0xC>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float rdlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// lane `lane` (wave-uniform) of the result takes `val` (wave-uniform), the other lanes keep `old`
// (what v_writelane_b32 does; written as a select because this hipcc has no writelane builtin)
__device__ __forceinline__ float wrlane(float val, int lane, float old) {
  return (int)__lane_id() == lane ? val : old;
}
__device__ __forceinline__ unsigned wrlane_u(unsigned val, int lane, unsigned old) {
  return (int)__lane_id() == lane ? val : old;
}

// lower bound of the canonical squared distance from p to any point of the box (see header)
__device__ __forceinline__ float box_lb(float lx, float ly, float lz, float hx, float hy, float hz, float px,
                                        float py, float pz) {
#pragma clang fp contract(off)
  const float gx = fmaxf(fmaxf(lx - px, px - hx), 0.0f);
  const float gy = fmaxf(fmaxf(ly - py, py - hy), 0.0f);
  const float gz = fmaxf(fmaxf(lz - pz, pz - hz), 0.0f);
  const float xx = gx * gx;
  const float yy = gy * gy;
  const float zz = gz * gz;
  return (xx + yy) + zz;
}

constexpr int MAX_CELLS = 2048;

struct GridSpec {
  float lox, loy, loz, invx, invy, invz;
  int gx, gy, gz;
};

__device__ __forceinline__ int cell_of(const GridSpec &g, float x, float y, float z) {
  int cx = (int)((x - g.lox) * g.invx), cy = (int)((y - g.loy) * g.invy), cz = (int)((z - g.loz) * g.invz);
  cx = min(max(cx, 0), g.gx - 1);
  cy = min(max(cy, 0), g.gy - 1);
  cz = min(max(cz, 0), g.gz - 1);
  if (cz & 1) cy = g.gy - 1 - cy;        // snake order: consecutive cells are always neighbours
  if ((cz * g.gy + cy) & 1) cx = g.gx - 1 - cx;
  return (cz * g.gy + cy) * g.gx + cx;
}

// NW waves, RPL rows per lane, MD_LDS: running minima in LDS (dynamic shared) or in the workspace
template <int NW, int RPL, bool MD_LDS>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(const float *__restrict__ xyz, float *__restrict__ ws,
                                                             int32_t *__restrict__ idx, int N, int m, int log2bs,
                                                             int nrows, size_t ws_stride) {
  constexpr int T = NW * 64;
  extern __shared__ __align__(16) unsigned char s_dyn[];
  __shared__ float s_red[2][NW][8];
  __shared__ float s_box[6];
  struct alignas(8) Entry { unsigned long long packed; float x, y, z, pad; };
  __shared__ Entry s_tab[2][NW];
  __shared__ unsigned long long s_best[3];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const float *P = xyz + (size_t)blockIdx.x * N * 3;
  int32_t *out = idx + (size_t)blockIdx.x * m;
  const int R64 = nrows * 64;
  float *sx = ws + (size_t)blockIdx.x * ws_stride;
  float *sy = sx + R64, *sz = sy + R64;
  int *sk = reinterpret_cast<int *>(sz + R64);
  float *mdg = MD_LDS ? reinterpret_cast<float *>(s_dyn) : reinterpret_cast<float *>(sk + R64);
  int *s_cnt = reinterpret_cast<int *>(s_dyn);  // [MAX_CELLS + 2], aliases the minima before they are initialised

  // ---------------- phase 1a: bounding box of the candidate points --------------------------------
  const float BIG = 3.0e38f;
  float lx = BIG, ly = BIG, lz = BIG, hx = -BIG, hy = -BIG, hz = -BIG;
  for (int k = t; k < N; k += T) {
    const float x = P[k * 3], y = P[k * 3 + 1], z = P[k * 3 + 2];
    if (!(sqnorm(x, y, z) < 0.001f)) {
      lx = fminf(lx, x); ly = fminf(ly, y); lz = fminf(lz, z);
      hx = fmaxf(hx, x); hy = fmaxf(hy, y); hz = fmaxf(hz, z);
    }
  }
  lx = wave_min_f32(lx); ly = wave_min_f32(ly); lz = wave_min_f32(lz);
  hx = wave_max_f32(hx); hy = wave_max_f32(hy); hz = wave_max_f32(hz);
  if (lane == 0) {
    s_red[0][wid][0] = lx; s_red[0][wid][1] = ly; s_red[0][wid][2] = lz;
    s_red[0][wid][3] = hx; s_red[0][wid][4] = hy; s_red[0][wid][5] = hz;
  }
  for (int c = t; c < MAX_CELLS + 2; c += T) s_cnt[c] = 0;
  __syncthreads();
  if (t < 6) {
    float v = s_red[0][0][t];
    for (int w = 1; w < NW; ++w) v = t < 3 ? fminf(v, s_red[0][w][t]) : fmaxf(v, s_red[0][w][t]);
    s_box[t] = v;
  }
  __syncthreads();
  GridSpec g;
  {
    g.lox = s_box[0]; g.loy = s_box[1]; g.loz = s_box[2];
    const float ex = fmaxf(s_box[3] - g.lox, 1e-6f), ey = fmaxf(s_box[4] - g.loy, 1e-6f),
                ez = fmaxf(s_box[5] - g.loz, 1e-6f);
    // ~32 points per cell, at most MAX_CELLS cells, cell edge proportional to the scene extents
    float target = fminf(fmaxf((float)N / 32.0f, 1.0f), (float)MAX_CELLS);
    const float e = cbrtf(ex * ey * ez / target);
    g.gx = min(max((int)(ex / e), 1), 64);
    g.gy = min(max((int)(ey / e), 1), 64);
    g.gz = min(max((int)(ez / e), 1), 64);
    while (g.gx * g.gy * g.gz > MAX_CELLS) {  // uniform decision: all threads compute the same values
      if (g.gx >= g.gy && g.gx >= g.gz) --g.gx; else if (g.gy >= g.gz) --g.gy; else --g.gz;
    }
    g.invx = (float)g.gx / ex; g.invy = (float)g.gy / ey; g.invz = (float)g.gz / ez;
  }
  const int ncell = g.gx * g.gy * g.gz;  // cell `ncell` collects the never-selectable points

  // ---------------- phase 1b: counting sort by cell ------------------------------------------------
  for (int k = t; k < N; k += T) {
    const float x = P[k * 3], y = P[k * 3 + 1], z = P[k * 3 + 2];
    const int c = (sqnorm(x, y, z) < 0.001f) ? ncell : cell_of(g, x, y, z);
    atomicAdd(&s_cnt[c], 1);
  }
  __syncthreads();
  if (wid == 0) {  // exclusive scan of <= MAX_CELLS+1 counters by one wave, 64 at a time
    int carry = 0;
    for (int base = 0; base <= ncell; base += 64) {
      const int c = base + lane;
      const int v = c <= ncell ? s_cnt[c] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (c <= ncell) s_cnt[c] = carry + incl - v;
      carry += __shfl(incl, 63);
    }
  }
  __syncthreads();
  for (int k = t; k < N; k += T) {
    const float x = P[k * 3], y = P[k * 3 + 1], z = P[k * 3 + 2];
    const int c = (sqnorm(x, y, z) < 0.001f) ? ncell : cell_of(g, x, y, z);
    const int pos = atomicAdd(&s_cnt[c], 1);
    sx[pos] = x; sy[pos] = y; sz[pos] = z; sk[pos] = k;
  }
  for (int pos = N + t; pos < R64; pos += T) {  // padding lanes of the last row
    sx[pos] = 0.0f; sy[pos] = 0.0f; sz[pos] = 0.0f; sk[pos] = 0;
  }
  __syncthreads();  // scratch complete (and visible workgroup-wide); the histogram is dead from here

  // ---------------- phase 1c: running minima + row records ------------------------------------------
  for (int pos = t; pos < R64; pos += T) {
    const bool cand = pos < N && !(sqnorm(sx[pos], sy[pos], sz[pos]) < 0.001f);
    mdg[pos] = cand ? 1e10f : -1.0f;  // -1: never a candidate (fminf(-1, d>=0) stays -1)
  }
  // record of row (slot*NW + wid) in lane slot%64, register set slot/64
  float blx[RPL], bly[RPL], blz[RPL], bhx[RPL], bhy[RPL], bhz[RPL], rmax[RPL], rbx[RPL], rby[RPL], rbz[RPL];
  unsigned rkey[RPL];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    blx[q] = bly[q] = blz[q] = BIG; bhx[q] = bhy[q] = bhz[q] = -BIG;
    rmax[q] = -1.0f; rkey[q] = 0xFFFFFFFFu; rbx[q] = rby[q] = rbz[q] = 0.0f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    for (int s = 0; s < 64; ++s) {
      const int row = (q * 64 + s) * NW + wid;
      if (row >= nrows) break;  // uniform
      const int pos = row * 64 + lane;
      const float x = sx[pos], y = sy[pos], z = sz[pos];
      const bool cand = mdg[pos] > 0.0f;
      const float a = wave_min_f32(cand ? x : BIG), b = wave_min_f32(cand ? y : BIG), c = wave_min_f32(cand ? z : BIG);
      const float d = wave_max_f32(cand ? x : -BIG), e = wave_max_f32(cand ? y : -BIG), f = wave_max_f32(cand ? z : -BIG);
      const bool any = __ballot(cand) != 0ull;
      blx[q] = wrlane(a, s, blx[q]); bly[q] = wrlane(b, s, bly[q]); blz[q] = wrlane(c, s, blz[q]);
      bhx[q] = wrlane(d, s, bhx[q]); bhy[q] = wrlane(e, s, bhy[q]); bhz[q] = wrlane(f, s, bhz[q]);
      rmax[q] = wrlane(any ? BIG : -1.0f, s, rmax[q]);  // BIG: forces a full first round
    }
  }

  // ---------------- phase 2: the sampling rounds -----------------------------------------------------
  float px = P[0], py = P[1], pz = P[2];
  IdSink sink{out, 0};
  sink.push(0, 0, m);
  if (t < 3) s_best[t] = 0ull;
  __syncthreads();
  int wbits = __float_as_int(-1.0f);  // this wave's cached best record (wave-uniform)
  unsigned wkey = 0xFFFFFFFFu;
  float wbx = 0.f, wby = 0.f, wbz = 0.f;
  int rot = 0;
  for (int j = 1; j < m; ++j) {
    bool changed = false;
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const float lb = box_lb(blx[q], bly[q], blz[q], bhx[q], bhy[q], bhz[q], px, py, pz);
      unsigned long long mask = __ballot(lb < rmax[q]);
      changed |= mask != 0ull;
      while (mask) {
        // two rows per trip so their loads and reduction chains overlap; with one row left the second slot
        // repeats the first (min/max are idempotent, the record is rewritten with the same values)
        const int s0 = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const int s1 = mask ? __builtin_ctzll(mask) : s0;
        mask &= mask - 1ull;  // no-op when mask is already 0
        const int p0 = ((q * 64 + s0) * NW + wid) * 64 + lane;
        const int p1 = ((q * 64 + s1) * NW + wid) * 64 + lane;
        const float x0 = sx[p0], y0 = sy[p0], z0 = sz[p0];
        const float x1 = sx[p1], y1 = sy[p1], z1 = sz[p1];
        const int k0 = sk[p0], k1 = sk[p1];
        const float n0 = fminf(mdg[p0], sqdist(x0, y0, z0, px, py, pz));
        const float n1 = fminf(mdg[p1], sqdist(x1, y1, z1, px, py, pz));
        mdg[p0] = n0;
        mdg[p1] = n1;
        int b0, b1;
        unsigned key0, key1;
        const int L0 = wave_argmax(n0, tie_key2(k0, log2bs), &b0, &key0);
        const int L1 = wave_argmax(n1, tie_key2(k1, log2bs), &b1, &key1);
        rmax[q] = wrlane(__int_as_float(b0), s0, rmax[q]);
        rkey[q] = wrlane_u(key0, s0, rkey[q]);
        rbx[q] = wrlane(rdlane(x0, L0), s0, rbx[q]);
        rby[q] = wrlane(rdlane(y0, L0), s0, rby[q]);
        rbz[q] = wrlane(rdlane(z0, L0), s0, rbz[q]);
        rmax[q] = wrlane(__int_as_float(b1), s1, rmax[q]);
        rkey[q] = wrlane_u(key1, s1, rkey[q]);
        rbx[q] = wrlane(rdlane(x1, L1), s1, rbx[q]);
        rby[q] = wrlane(rdlane(y1, L1), s1, rby[q]);
        rbz[q] = wrlane(rdlane(z1, L1), s1, rbz[q]);
      }
    }
    if (changed) {  // wave-uniform: refresh the wave's best over its row records
      float v = rmax[0];
      unsigned kq = rkey[0];
      float bx = rbx[0], by = rby[0], bz = rbz[0];
#pragma unroll
      for (int q = 1; q < RPL; ++q) {
        const bool better = rmax[q] > v || (rmax[q] == v && rkey[q] < kq);
        v = better ? rmax[q] : v; kq = better ? rkey[q] : kq;
        bx = better ? rbx[q] : bx; by = better ? rby[q] : by; bz = better ? rbz[q] : bz;
      }
      const int L = wave_argmax(v, kq, &wbits, &wkey);
      wbx = rdlane(bx, L); wby = rdlane(by, L); wbz = rdlane(bz, L);
    }
    const int buf = j & 1;
    const unsigned long long mine = pack_best(wbits, wkey);
    if (lane == 0) {
      s_tab[buf][wid].packed = mine;
      s_tab[buf][wid].x = wbx; s_tab[buf][wid].y = wby; s_tab[buf][wid].z = wbz;
      atomicMax(&s_best[rot], mine);
    }
    barrier_lds_only();  // global traffic of a round (row loads / minima) is private to the issuing lane
    {
      const unsigned long long win = s_best[rot];
      if (t == 0) s_best[rot == 0 ? 2 : rot - 1] = 0ull;
      rot = rot == 2 ? 0 : rot + 1;
      int old;
      if ((win >> 32) == 0ull) {  // no candidate anywhere: the reference's (-1, 0) fallback
        old = 0;
        px = P[0]; py = P[1]; pz = P[2];
      } else {
        const int src = lane < NW ? lane : 0;
        const unsigned long long e = s_tab[buf][src].packed;
        const float ex = s_tab[buf][src].x, ey = s_tab[buf][src].y, ez = s_tab[buf][src].z;
        const int L = __builtin_ctzll(__ballot(lane < NW && e == win) | (1ull << 63));
        old = (int)((~(unsigned)win) & 0x3FFFFFu);
        px = rdlane(ex, L); py = rdlane(ey, L); pz = rdlane(ez, L);
      }
      sink.push(j, old, m);
    }
  }
}

}  // namespace bq

using namespace bq;

namespace {
constexpr int BUCKET_NW = 16;
constexpr size_t LDS_BUDGET = 160 * 1024 - 2048;  // dynamic part; static reductions + slack stay below 2 KB

inline int rows_of(int N) { return (N + 63) / 64; }
}  // namespace

// bytes of scratch bq_furthest_point_sampling needs for (B, N); 0 when the register kernel is used
extern "C" size_t bq_fps_workspace_bytes(int B, int N) {
  if (N <= 4096 || B <= 0) return 0;
  const size_t r64 = (size_t)rows_of(N) * 64;
  return (size_t)B * r64 * 5 * sizeof(float);  // sx, sy, sz, sk (+ md when it does not fit in LDS)
}

namespace bq {

int launch_fps_bucket(const float *xyz, void *workspace, size_t workspace_bytes, int32_t *idx, int B, int N, int m,
                      int log2bs, hipStream_t st) {
  BQ_REQUIRE(workspace && workspace_bytes >= bq_fps_workspace_bytes(B, N), BQ_EINVAL,
             "fps: workspace of %zu bytes required for B=%d N=%d", bq_fps_workspace_bytes(B, N), B, N);
  const int nrows = rows_of(N);
  const size_t r64 = (size_t)nrows * 64;
  const size_t stride = r64 * 5;
  const int slots = (nrows + BUCKET_NW - 1) / BUCKET_NW;  // rows per wave
  const int rpl = (slots + 63) / 64;
  const size_t md_bytes = r64 * sizeof(float);
  const size_t hist_bytes = (MAX_CELLS + 2) * sizeof(int);
  const bool md_lds = md_bytes <= LDS_BUDGET;
  const size_t dyn = md_lds ? (md_bytes > hist_bytes ? md_bytes : hist_bytes) : hist_bytes;
  float *ws = reinterpret_cast<float *>(workspace);
#define BQ_LAUNCH_BUCKET(RPL, MDL)                                                                              \
  do {                                                                                                          \
    auto kern = fps_bucket_kernel<BUCKET_NW, RPL, MDL>;                                                         \
    static bool lds_reserved = false; /* once per instantiation: keeps the launch path free of driver calls */  \
    if (!lds_reserved) {                                                                                        \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BUDGET);          \
      if (e != hipSuccess) { set_error("fps: cannot reserve LDS: %s", hipGetErrorString(e)); return (int)e; }   \
      lds_reserved = true;                                                                                      \
    }                                                                                                           \
    hipLaunchKernelGGL(kern, dim3(B), dim3(BUCKET_NW * 64), dyn, st, xyz, ws, idx, N, m, log2bs, nrows, stride); \
  } while (0)
  if (rpl == 1) { if (md_lds) BQ_LAUNCH_BUCKET(1, true); else BQ_LAUNCH_BUCKET(1, false); }
  else if (rpl == 2) BQ_LAUNCH_BUCKET(2, false);
  else if (rpl <= 4) BQ_LAUNCH_BUCKET(4, false);
  else if (rpl <= 8) BQ_LAUNCH_BUCKET(8, false);
  else { set_error("fps: N=%d too large for the bucketed kernel", N); return BQ_ELIMIT; }
#undef BQ_LAUNCH_BUCKET
  return check_launch("furthest_point_sampling(bucketed)");
}

}  // namespace bq
