// Deterministic scatter gradients through an INVERTED INDEX.
//
// group_points_grad / three_interpolate_grad of the reference (lib/pointnet2/_ext_src/src/group_points_gpu.cu:43-75,
// interpolate_gpu.cu:116-154) scatter with fp32 atomics: the order of the additions -- and with it the rounding of every sum --
// changes from launch to launch.  Those sums feed the deepest layers of the detector; two executions of the same step
// differed by 2.6-7.9 % on BatchNorm bias gradients that are sums of nearly cancelling terms.  Here the scatter is turned
// around: bq_invert_index sorts the positions of an index tensor by the point they name (a STABLE LSD radix sort of
// (scene * N + point, position) pairs, 8 bits per pass -- within a point the positions stay ascending) into a CSR table, and
// the gradient kernels GATHER: one sum per destination, its terms added in ascending position order.  Same terms as the
// reference, a fixed order, no atomics, no zero-fill; bitwise reproducible.
// (The sort is this file's own: rocPRIM's DeviceRadixSort, the first version, ran eagerly at every size and replayed from a
// HIP graph at c3's 2.1 M pairs, but a REPLAYED graph holding it at c5's 4.2 M pairs died with a memory access fault (bisected
// in round 5); everything here is plain kernels on caller-provided scratch.)

#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

__global__ __launch_bounds__(256) void invert_keys_kernel(const int32_t *__restrict__ idx, unsigned *__restrict__ keys,
                                                          unsigned *__restrict__ vals, long L, int N, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / L;
    int v = idx[i];
    v = v < 0 ? 0 : (v >= N ? N - 1 : v);   // (the operators clamp nothing; an index outside [0, N) is the caller's bug)
    keys[i] = (unsigned)(b * N + v);
    vals[i] = (unsigned)i;
  }
}

// start[k] = first sorted position whose key is >= k, for k = 0 .. K (K = B * N): entry i fills the keys it opens
__global__ __launch_bounds__(256) void invert_starts_kernel(const unsigned *__restrict__ keys, int32_t *__restrict__ start,
                                                            long total, long K) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i <= total; i += (long)gridDim.x * 256) {
    const long lo = i == 0 ? 0 : (long)keys[i - 1] + 1;
    const long hi = i == total ? K : (long)keys[i];
    for (long k = lo; k <= hi; ++k) start[k] = (int32_t)i;
  }
}

// grad_feats[b][n][c] = sum over the positions p = (b, m, s) with idx[p] == n, ascending, of grad_out[p][3 + c].
// grad_out rows of `ld` elements (bf16 or fp32), the three offset channels first (group_concat_pm's layout).
// LP lanes per point, one 8-element chunk of the row per lane and trip (chunk j covers channels 8 j - 3 .. 8 j + 4).
// grad_xyz (optional): f32 (B, N, 3) = the same sum over channels 0..2, times xyz_scale (1 / radius under normalize_xyz) --
// the grouped offsets are (xyz[idx] - centre) / radius, so the point receives +g, its centre -g (kernel below).
template <typename OT, int LP>
__global__ __launch_bounds__(256) void group_concat_pm_grad_gather_kernel(const OT *__restrict__ grad_out,
                                                                          const int32_t *__restrict__ start,
                                                                          const unsigned *__restrict__ slots,
                                                                          float *__restrict__ grad_feats, int C, long points, int ld,
                                                                          float *__restrict__ grad_xyz, float xyz_scale) {
  const int sub = threadIdx.x % LP;
  const int chunks = ld >> 3;
  for (long pt = ((long)blockIdx.x * 256 + threadIdx.x) / LP; pt < points; pt += (long)gridDim.x * 256 / LP) {
    const int e0 = start[pt], e1 = start[pt + 1];
    for (int j = sub; j < chunks; j += LP) {
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
      for (int e = e0; e < e1; ++e) {
        const OT *g = grad_out + (long)slots[e] * ld + j * 8;
        if constexpr (sizeof(OT) == 2) {
          const uint4 raw = *reinterpret_cast<const uint4 *>(g);
          const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[2 * q] += __uint_as_float(w[q] << 16);
            acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
          }
        } else {
          const float4 a = *reinterpret_cast<const float4 *>(g), b = *reinterpret_cast<const float4 *>(g + 4);
          acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
          acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
      }
      if (grad_feats) {
        float *o = grad_feats + pt * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = j * 8 + e - 3;
          if (c >= 0 && c < C) o[c] = acc[e];
        }
      }
      if (grad_xyz && j == 0) {
        grad_xyz[pt * 3 + 0] = acc[0] * xyz_scale;
        grad_xyz[pt * 3 + 1] = acc[1] * xyz_scale;
        grad_xyz[pt * 3 + 2] = acc[2] * xyz_scale;
      }
    }
  }
}

// grad_new_xyz[b][m][:] = - scale * sum_s grad_out[b][m][s][0..2], s ascending: a centre's own S rows, no collision to begin with
template <typename OT>
__global__ __launch_bounds__(256) void group_concat_pm_grad_centres_kernel(const OT *__restrict__ grad_out, float *__restrict__ grad_new_xyz,
                                                                           long centres, int S, int ld, float scale) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < centres * 3; t += (long)gridDim.x * 256) {
    const long cm = t / 3;
    const int a = (int)(t - cm * 3);
    float acc = 0.f;
    for (int sidx = 0; sidx < S; ++sidx) acc += (float)grad_out[(cm * S + sidx) * ld + a];
    grad_new_xyz[t] = -acc * scale;
  }
}

// grad_points[b][c][m] = sum over the positions p = (b, j, k) with idx[p] == m, ascending, of grad_out[b][c][j] * weight[b][j][k]
__global__ __launch_bounds__(256) void three_interpolate_grad_gather_kernel(const float *__restrict__ grad_out,
                                                                            const int32_t *__restrict__ start,
                                                                            const unsigned *__restrict__ slots,
                                                                            const float *__restrict__ weight,
                                                                            float *__restrict__ grad_points, int C, int n, int m,
                                                                            long total) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int mi = (int)(t % m);
    const long bc = t / m;
    const int c = (int)(bc % C);
    const long b = bc / C;
    const int e0 = start[b * m + mi], e1 = start[b * m + mi + 1];
    const float *go = grad_out + (b * C + c) * (long)n;
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) {
      const unsigned p = slots[e];             // = (b * n + j) * 3 + k
      const long j = (long)(p / 3u) - b * n;
      acc += go[j] * weight[p];
    }
    grad_points[t] = acc;
  }
}

}  // namespace bq

namespace bq {

// ---- stable LSD radix sort of (key, value) pairs, 8 bits per pass ------------------------------------------------------------
// block = 256 threads x RS_IPT items in position order.  hist: per-block digit counts, stored digit-major [256][nblocks];
// scan: one workgroup turns them into global offsets (exclusive, digit-major order = the sorted order of (digit, block));
// scatter: a block walks its items in rounds of 256 (thread t takes item r * 256 + t): rank among equal digits = items of
// earlier rounds (running counters in LDS) + earlier waves of this round (per-wave counts in LDS) + lower lanes of this wave
// (eight ballots give the lanes with the same digit) -- every tie broken by position, so the sort is stable.
constexpr int RS_IPT = 8, RS_ITEMS = 256 * RS_IPT;

// digit of a radix pass.  A pass at shift >= 32 exists when the pass count was rounded up to an odd number for 25 .. 32 key
// bits (K >= 2^24): a 32-bit shift by 32 is undefined (the hardware shifts by 0 and the pass would re-sort by the low byte) --
// such a pass sees digit 0 everywhere, i.e. it is the stable copy it is meant to be.
__device__ __forceinline__ unsigned rs_digit(unsigned key, int shift) { return shift >= 32 ? 0u : (key >> shift) & 255u; }

__global__ __launch_bounds__(256) void rs_hist_kernel(const unsigned *__restrict__ keys, int *__restrict__ hist, long total, int nblocks,
                                                      int shift) {
  __shared__ int s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const long base = (long)blockIdx.x * RS_ITEMS;
  for (int r = 0; r < RS_IPT; ++r) {
    const long i = base + r * 256 + threadIdx.x;
    if (i < total) atomicAdd(&s_h[rs_digit(keys[i], shift)], 1);
  }
  __syncthreads();
  hist[(long)threadIdx.x * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's row of block counts in place, the row's total to tot[digit]
__global__ __launch_bounds__(256) void rs_scan_kernel(int *__restrict__ hist, int *__restrict__ tot, int nblocks) {
  __shared__ int s_w[4];
  __shared__ int s_carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int *a = hist + (long)blockIdx.x * nblocks;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    int v[4], sum = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = base + t * 4 + e;
      v[e] = i < nblocks ? a[i] : 0;
      sum += v[e];
    }
    int incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int x = 0; x < w; ++x) wbase += s_w[x];
    int run = s_carry + wbase + incl - sum;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = base + t * 4 + e;
      if (i < nblocks) a[i] = run;
      run += v[e];
    }
    __syncthreads();
    if (t == 255) s_carry = s_carry + wbase + incl;
    __syncthreads();
  }
  if (t == 0) tot[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const unsigned *__restrict__ keys_in, const unsigned *__restrict__ vals_in,
                                                         unsigned *__restrict__ keys_out, unsigned *__restrict__ vals_out,
                                                         const int *__restrict__ offs, const int *__restrict__ tot, long total,
                                                         int nblocks, int shift) {
  __shared__ int s_run[256];       // where this block's next item of a digit goes
  __shared__ int s_cnt[4][256];    // this round's count per (wave, digit)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  {  // digit base = exclusive scan of the 256 row totals
    const int mine = tot[t];
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_cnt[0][w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int x = 0; x < w; ++x) wbase += s_cnt[0][x];
    s_run[t] = wbase + incl - mine + offs[(long)t * nblocks + blockIdx.x];
    __syncthreads();
  }
  const long base = (long)blockIdx.x * RS_ITEMS;
  for (int r = 0; r < RS_IPT; ++r) {
#pragma unroll
    for (int x = 0; x < 4; ++x) s_cnt[x][t] = 0;
    __syncthreads();
    const long i = base + r * 256 + t;
    const bool live = i < total;
    const unsigned k = live ? keys_in[i] : 0u;
    const unsigned d = rs_digit(k, shift);
    // lanes of this wave with the same digit (dead lanes form their own class through the `live` ballot)
    unsigned long long peers = __ballot(live);
    peers = live ? peers : ~peers;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const int below = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
    if (live && below == 0) s_cnt[w][d] = __builtin_popcountll(peers);   // the class's first lane
    __syncthreads();
    if (live) {
      int pos = s_run[d] + below;
      for (int x = 0; x < w; ++x) pos += s_cnt[x][d];
      keys_out[pos] = k;
      vals_out[pos] = vals_in[i];
    }
    __syncthreads();
    s_run[t] += s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
    __syncthreads();
  }
}

}  // namespace bq

// scratch: two key buffers + one value buffer (the second value buffer is `slots`) + the block histograms
extern "C" size_t bq_invert_index_workspace_bytes(long total) {
  if (total <= 0) return 0;
  const long nblocks = (total + bq::RS_ITEMS - 1) / bq::RS_ITEMS;
  return 3 * (size_t)total * sizeof(unsigned) + (size_t)256 * (nblocks + 1) * sizeof(int);
}

// idx int32 (B, L) with values in [0, N): -> start int32 [B * N + 1] (CSR over scene * N + value), slots uint32 [B * L] = the
// positions b * L + l of every value in ascending order.  workspace: bq_invert_index_workspace_bytes(B * L) bytes.
extern "C" int bq_invert_index(const int32_t *idx, int B, long L, int N, int32_t *start, unsigned *slots, void *workspace,
                               size_t workspace_bytes, void *stream) {
  using namespace bq;
  BQ_REQUIRE(B >= 0 && L >= 0 && N > 0, BQ_EINVAL, "invert_index: bad extents");
  const long total = (long)B * L, K = (long)B * N;
  BQ_REQUIRE(total < (1L << 31) - RS_ITEMS && K < (1L << 31), BQ_ELIMIT, "invert_index: more than 2^31 entries");
  BQ_REQUIRE(start, BQ_EINVAL, "invert_index: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (total == 0) {
    hipLaunchKernelGGL(invert_starts_kernel, dim3(1), dim3(256), 0, st, (const unsigned *)nullptr, start, 0L, K);
    return check_launch("invert_index");
  }
  BQ_REQUIRE(idx && slots && workspace && workspace_bytes >= bq_invert_index_workspace_bytes(total), BQ_EINVAL,
             "invert_index: workspace of %zu bytes required", bq_invert_index_workspace_bytes(total));
  const int nblocks = (int)((total + RS_ITEMS - 1) / RS_ITEMS);
  unsigned *ka = (unsigned *)workspace, *kb = ka + total, *va = kb + total;
  int *hist = (int *)(va + total), *tot = hist + 256L * nblocks;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(invert_keys_kernel, dim3(blocks), dim3(256), 0, st, idx, ka, va, L, N, total);
  int bits = 1;
  while ((1L << bits) < K) ++bits;
  int passes = (bits + 7) / 8;
  if ((passes & 1) == 0) ++passes;   // an odd number of passes: the values end in `slots` (a pass over zero bits is a stable copy)
  unsigned *kin = ka, *kout = kb, *vin = va, *vout = slots;
  for (int p = 0; p < passes; ++p) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nblocks), dim3(256), 0, st, (const unsigned *)kin, hist, total, nblocks, 8 * p);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, st, hist, tot, nblocks);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nblocks), dim3(256), 0, st, (const unsigned *)kin, (const unsigned *)vin, kout, vout,
                       (const int *)hist, (const int *)tot, total, nblocks, 8 * p);
    unsigned *tk = kin; kin = kout; kout = tk;
    unsigned *tv = vin; vin = vout; vout = tv;
  }
  // after an odd number of passes the sorted keys are in kb and the values in slots
  hipLaunchKernelGGL(invert_starts_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned *)kin, start, total, K);
  return check_launch("invert_index");
}

extern "C" int bq_group_concat_pm_grad_gather(const void *grad_out, int in_bf16, const int32_t *start, const unsigned *slots,
                                              float *grad_feats, float *grad_xyz, float *grad_new_xyz, int B, int C, int N, int M,
                                              int S, int ld, float radius, int normalize, void *stream) {
  using namespace bq;
  BQ_REQUIRE(grad_out && start && slots && (grad_feats || grad_xyz || grad_new_xyz), BQ_EINVAL, "group_concat_pm_grad_gather: null pointer");
  BQ_REQUIRE(B >= 0 && C >= 0 && N > 0 && ld >= C + 3 && ld % 8 == 0, BQ_EINVAL, "group_concat_pm_grad_gather: ld = %d must be a multiple of 8 >= 3 + C", ld);
  BQ_REQUIRE(!grad_feats || C > 0, BQ_EINVAL, "group_concat_pm_grad_gather: no feature channels");
  BQ_REQUIRE(!grad_new_xyz || (M > 0 && S > 0), BQ_EINVAL, "group_concat_pm_grad_gather: grad_new_xyz needs M, S");
  const float xs = normalize ? 1.0f / radius : 1.0f;
  BQ_REQUIRE(((uintptr_t)grad_out % 16) == 0, BQ_EINVAL, "group_concat_pm_grad_gather: grad_out must be 16-byte aligned");
  const long points = (long)B * N;
  if (points == 0) return BQ_OK;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = ld / 8;
  long blocks;
  if (chunks <= 16) {
    blocks = (points * 16 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (in_bf16) hipLaunchKernelGGL((group_concat_pm_grad_gather_kernel<__bf16, 16>), dim3((unsigned)blocks), dim3(256), 0, st,
                                    (const __bf16 *)grad_out, start, slots, grad_feats, C, points, ld, grad_xyz, xs);
    else hipLaunchKernelGGL((group_concat_pm_grad_gather_kernel<float, 16>), dim3((unsigned)blocks), dim3(256), 0, st,
                            (const float *)grad_out, start, slots, grad_feats, C, points, ld, grad_xyz, xs);
  } else {
    blocks = (points * 32 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (in_bf16) hipLaunchKernelGGL((group_concat_pm_grad_gather_kernel<__bf16, 32>), dim3((unsigned)blocks), dim3(256), 0, st,
                                    (const __bf16 *)grad_out, start, slots, grad_feats, C, points, ld, grad_xyz, xs);
    else hipLaunchKernelGGL((group_concat_pm_grad_gather_kernel<float, 32>), dim3((unsigned)blocks), dim3(256), 0, st,
                            (const float *)grad_out, start, slots, grad_feats, C, points, ld, grad_xyz, xs);
  }
  if (grad_new_xyz) {
    const long centres = (long)B * M;
    long cb = (centres * 3 + 255) / 256;
    if (cb > 65536) cb = 65536;
    if (in_bf16) hipLaunchKernelGGL(group_concat_pm_grad_centres_kernel<__bf16>, dim3((unsigned)cb), dim3(256), 0, st,
                                    (const __bf16 *)grad_out, grad_new_xyz, centres, S, ld, xs);
    else hipLaunchKernelGGL(group_concat_pm_grad_centres_kernel<float>, dim3((unsigned)cb), dim3(256), 0, st, (const float *)grad_out,
                            grad_new_xyz, centres, S, ld, xs);
  }
  return check_launch("group_concat_pm_grad_gather");
}

extern "C" int bq_three_interpolate_grad_gather(const float *grad_out, const int32_t *start, const unsigned *slots,
                                                const float *weight, float *grad_points, int B, int C, int n, int m, void *stream) {
  using namespace bq;
  BQ_REQUIRE(grad_out && start && slots && weight && grad_points, BQ_EINVAL, "three_interpolate_grad_gather: null pointer");
  BQ_REQUIRE(B >= 0 && C >= 0 && n >= 0 && m > 0, BQ_EINVAL, "three_interpolate_grad_gather: bad extents");
  const long total = (long)B * C * m;
  if (total == 0) return BQ_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(three_interpolate_grad_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_out, start,
                     slots, weight, grad_points, C, n, m, total);
  return check_launch("three_interpolate_grad_gather");
}
