// Post-processing of the detector's proposals on the device -- SURVEY.md §8f rank 3 (reference lib/ap_helper.py:40-178
// parse_predictions over utils/nms.py:40-152 and utils/box_util.py:282-300): the reference decodes B x K boxes in a Python
// double loop on the host (one device->host sync per scalar), tests every box against all N points with a scipy Delaunay
// hull (remove_empty_box) and runs a numpy greedy NMS per scene.  Here: one kernel counts the points inside every box,
// one kernel per launch does the greedy NMS of all scenes (one workgroup per scene), both on device tensors.
#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

// ---- points inside an oriented box ------------------------------------------------------------------------------------
// box (b, k): centre c, size (l, w, h), heading t; corners = roty(t) . local + c (box_util.py:282-300), so a point p is inside
// iff local = roty(t)^T (p - c) has |x| <= l/2, |y| <= w/2, |z| <= h/2 (the closed box = the convex hull of the corners,
// which is what the reference's in-hull test decides).  One workgroup per (box, scene); the scene's points (xyz at the
// head of rows of `ld` floats) stream through; per-wave ballot counts.
__global__ __launch_bounds__(256) void box_point_count_kernel(const float *__restrict__ pc, const float *__restrict__ center,
                                                             const float *__restrict__ size, const float *__restrict__ heading,
                                                             int *__restrict__ count, int N, int ld, int K, int cap) {
  __shared__ int s_cnt[4];
  const int k = blockIdx.x, b = blockIdx.y;
  const float *c = center + ((long)b * K + k) * 3, *sz = size + ((long)b * K + k) * 3;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float hl = 0.5f * sz[0], hw = 0.5f * sz[1], hh = 0.5f * sz[2];
  const float t = heading[(long)b * K + k], ct = cosf(t), st = sinf(t);
  const float *p = pc + (long)b * N * ld;
  int mine = 0;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float dx = p[(long)i * ld] - cx, dy = p[(long)i * ld + 1] - cy, dz = p[(long)i * ld + 2] - cz;
    const float x = ct * dx - st * dz, z = st * dx + ct * dz;
    mine += (fabsf(x) <= hl && fabsf(dy) <= hw && fabsf(z) <= hh) ? 1 : 0;
    // (cap: the caller only asks "fewer than cap?" -- ap_helper.py:98 `< 5` -- but a wave-uniform early exit would
    // cost a ballot per iteration; N = 40000 is 157 iterations)
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    count[(long)b * K + k] = cap > 0 && tot > cap ? cap : tot;
  }
}

// ---- greedy NMS, one workgroup per scene ----------------------------------------------------------------------------------
// utils/nms.py:40-152 (nms_2d_faster / nms_3d_faster / nms_3d_faster_samecls): visit the boxes in DEcreasing score order;
// a visited box that is still alive is picked and suppresses every other alive box whose overlap with it exceeds the
// threshold -- overlap = IoU, or (old_type) intersection / the OTHER box's volume; with same_cls only boxes of the picked
// box's class can be suppressed.  Boxes with valid == 0 (empty boxes removed beforehand) do not take part.
// box: (B, K, 6) = (x1, y1, z1, x2, y2, z2); for the 2-D variant the caller passes z1 = 0, z2 = 1.  Arithmetic in double,
// as numpy's.  Ties of the score are visited in increasing index order (numpy's argsort order for ties is unspecified).
constexpr int NMS_MAX_K = 1024;

__global__ __launch_bounds__(256) void nms_kernel(const float *__restrict__ box, const float *__restrict__ score,
                                                  const int *__restrict__ cls, const unsigned char *__restrict__ valid,
                                                  unsigned char *__restrict__ keep, int K, float thresh, int old_type,
                                                  int same_cls) {
  __shared__ float s_score[NMS_MAX_K];
  __shared__ short s_order[NMS_MAX_K];
  __shared__ unsigned char s_alive[NMS_MAX_K];
  __shared__ int s_pick;
  const int b = blockIdx.x, t = threadIdx.x;
  const float *bx = box + (long)b * K * 6;
  for (int i = t; i < K; i += 256) {
    s_score[i] = score[(long)b * K + i];
    s_alive[i] = valid ? valid[(long)b * K + i] : 1;
    keep[(long)b * K + i] = 0;
  }
  __syncthreads();
  // rank sort (K <= 1024: K^2 / 256 comparisons per thread): position of i in (score desc, index asc)
  for (int i = t; i < K; i += 256) {
    const float si = s_score[i];
    int rank = 0;
    for (int j = 0; j < K; ++j) {
      const float sj = s_score[j];
      rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
    }
    s_order[rank] = (short)i;
  }
  __syncthreads();
  for (int pos = 0; pos < K; ++pos) {
    if (t == 0) s_pick = s_alive[s_order[pos]] ? s_order[pos] : -1;
    __syncthreads();
    const int i = s_pick;
    if (i >= 0) {
      const double x1 = bx[i * 6], y1 = bx[i * 6 + 1], z1 = bx[i * 6 + 2], x2 = bx[i * 6 + 3], y2 = bx[i * 6 + 4],
                   z2 = bx[i * 6 + 5];
      const double vi = (x2 - x1) * (y2 - y1) * (z2 - z1);
      const int ci = (same_cls && cls) ? cls[(long)b * K + i] : 0;
      for (int j = t; j < K; j += 256) {
        if (!s_alive[j] || j == i) continue;
        const double a1 = bx[j * 6], b1 = bx[j * 6 + 1], c1 = bx[j * 6 + 2], a2 = bx[j * 6 + 3], b2 = bx[j * 6 + 4],
                     c2 = bx[j * 6 + 5];
        const double l = fmax(0.0, fmin(x2, a2) - fmax(x1, a1)), w = fmax(0.0, fmin(y2, b2) - fmax(y1, b1)),
                     h = fmax(0.0, fmin(z2, c2) - fmax(z1, c1));
        const double inter = l * w * h, vj = (a2 - a1) * (b2 - b1) * (c2 - c1);
        double o = old_type ? inter / vj : inter / (vi + vj - inter);
        if (same_cls && cls && cls[(long)b * K + j] != ci) o = 0.0;
        if (o > (double)thresh) s_alive[j] = 0;
      }
      if (t == 0) {
        keep[(long)b * K + i] = 1;
        s_alive[i] = 0;
      }
    }
    __syncthreads();
  }
}

}  // namespace bq

using namespace bq;

extern "C" __attribute__((visibility("default"))) int bq_box_point_count(
    const float *points, const float *center, const float *size, const float *heading, int *count, int B, int N, int ld,
    int K, int cap, void *stream) {
  BQ_REQUIRE(points && center && size && heading && count, BQ_EINVAL, "box_point_count: null pointer");
  BQ_REQUIRE(B > 0 && N > 0 && K > 0 && ld >= 3, BQ_EINVAL, "box_point_count: bad extents");
  hipLaunchKernelGGL(box_point_count_kernel, dim3(K, B), dim3(256), 0, (hipStream_t)stream, points, center, size, heading,
                     count, N, ld, K, cap);
  return check_launch("box_point_count");
}

extern "C" __attribute__((visibility("default"))) int bq_nms(
    const float *box, const float *score, const int *cls, const unsigned char *valid, unsigned char *keep, int B, int K,
    float thresh, int old_type, int same_cls, void *stream) {
  BQ_REQUIRE(box && score && keep, BQ_EINVAL, "nms: null pointer");
  BQ_REQUIRE(B > 0 && K > 0 && K <= NMS_MAX_K, BQ_ELIMIT, "nms: K = %d outside 1..%d", K, NMS_MAX_K);
  BQ_REQUIRE(!same_cls || cls, BQ_EINVAL, "nms: same_cls needs the class ids");
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, box, score, cls, valid, keep, K, thresh,
                     old_type, same_cls);
  return check_launch("nms");
}
