// Multiview projection of image features onto the scene's points -- SURVEY.md §8f rank 4 (reference lib/projection.py:5-276
// ProjectionHelper.compute_projection / project, driven per frame by scripts/project_multiview_features.py:103-202): the
// reference decides, frame by frame, which points a camera sees (frustum planes, pinhole projection, depth agreement) with a
// chain of boolean-mask compactions -- each a device->host synchronisation -- and then max-pools / first-fills the frames'
// features into the points one frame at a time.  Here: ONE launch maps every (frame, point) pair to its pixel (or -1) and
// ONE launch walks the frames of every point, both streaming (HBM-bound integer / gather work, no LDS, no MFMA).
#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

struct ProjFrame {       // per frame, precomputed by the host exactly as ProjectionHelper does (projection.py:58-130, :212)
  float w2c[16];         // torch.inverse(camera_to_world), row-major
  float normal[6][3];    // inward plane normals of the viewing frustum (compute_frustum_normals)
  float p1[3], p2[3];    // corner 2 and corner 4 of the frustum: a point of planes 0-2 / of planes 3-5
};

// pix[f][n] = y * W + x of the pixel point n projects to in frame f, or -1 when the frame does not see it:
// projection.py:138-156 (inside all six planes: round(100 d) / 100 < 0 with d = (p - plane point) . normal), :226-231
// (camera = w2c . (p, 1); u = x fx / z + cx, v = y fy / z + cy; rounded half-to-even), :234-236 (inside the image), :241-242
// (depth_min <= depth[pixel] <= depth_max and |depth[pixel] - z| <= accuracy).  fp32 throughout, multiply-adds in the
// order of a row-times-column product; the build runs with -ffp-contract=off, so only the explicit fmaf calls fuse.
__global__ __launch_bounds__(256) void project_points_kernel(const float *__restrict__ points, const float *__restrict__ depth,
                                                             const ProjFrame *__restrict__ frames, int *__restrict__ pix,
                                                             int N, int W, int H, float fx, float fy, float cx, float cy,
                                                             float dmin, float dmax, float acc) {
  __shared__ ProjFrame fr;
  const int f = blockIdx.y;
  if (threadIdx.x < sizeof(ProjFrame) / 4) reinterpret_cast<float *>(&fr)[threadIdx.x] = reinterpret_cast<const float *>(frames + f)[threadIdx.x];
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float x = points[3 * n], y = points[3 * n + 1], z = points[3 * n + 2];
  bool in = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float *o = k < 3 ? fr.p1 : fr.p2;
    const float dx = x - o[0], dy = y - o[1], dz = z - o[2];
    const float d = fmaf(dz, fr.normal[k][2], fmaf(dy, fr.normal[k][1], dx * fr.normal[k][0]));
    in = in && (rintf(d * 100.0f) / 100.0f < 0.0f);
  }
  int out = -1;
  if (in) {
    float c[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      c[r] = fmaf(fr.w2c[4 * r + 3], 1.0f, fmaf(fr.w2c[4 * r + 2], z, fmaf(fr.w2c[4 * r + 1], y, fr.w2c[4 * r] * x)));
    const float u = (c[0] * fx) / c[2] + cx, v = (c[1] * fy) / c[2] + cy;
    const float ru = rintf(u), rv = rintf(v);
    if (ru >= 0.0f && rv >= 0.0f && ru < (float)W && rv < (float)H) {
      const int p = (int)rv * W + (int)ru;
      const float dv = depth[(long)f * W * H + p];
      if (dv >= dmin && dv <= dmax && fabsf(dv - c[2]) <= acc) out = p;
    }
  }
  pix[(long)f * N + n] = out;
}

// point_features[n][:] from the frames in order (scripts/project_multiview_features.py:171-198), one wave per point, a
// lane per CPL channels (and per frame while the map is read); feat is pixel-major (F, H*W, C): a covered (frame, point) pair is one contiguous C-float row.
//   first-fill (maxpool = 0): a point whose vector is still all zero takes the frame's vector (:192-197);
//   max-pool   (maxpool = 1): a frame's vector that is not all zero fills an all-zero point and is max-ed into a filled
//   one (:179-191) -- "all zero" re-evaluated at every frame, as the script does (a max of negative entries against zeros
//   can empty a point again).
template <int CPL>
__global__ __launch_bounds__(1024) void fuse_point_features_kernel(const int *__restrict__ pix, const float *__restrict__ feat,
                                                                   float *__restrict__ out, int F, int N, int HW,
                                                                   int maxpool) {
  constexpr int C = 64 * CPL;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 16 + (threadIdx.x >> 6);   // 16 consecutive points per workgroup: their map entries share
  if (n >= N) return;                                   // 64-B sectors of every frame's row
  float cur[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) cur[j] = 0.0f;
  bool cur_zero = true;
  for (int f0 = 0; f0 < F; f0 += 64) {
    // the map entries of 64 frames in one load (lane = frame); the frames that see the point, in order, from the ballot
    const int mine = f0 + lane < F ? pix[(long)(f0 + lane) * N + n] : -1;
    unsigned long long seen = __ballot(mine >= 0);
    while (seen) {
      const int fl = __builtin_ctzll(seen);
      seen &= seen - 1;
      const int p = __builtin_amdgcn_readlane(mine, fl);
      const float *row = feat + ((long)(f0 + fl) * HW + p) * C + lane * CPL;
      float v[CPL];
      bool nz = false;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        v[j] = row[j];
        nz = nz || v[j] != 0.0f;
      }
      const bool new_nz = __ballot(nz) != 0ull;
      if (maxpool ? !new_nz : !cur_zero) continue;
      bool cnz = false;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        cur[j] = (cur_zero || !maxpool) ? v[j] : fmaxf(cur[j], v[j]);
        cnz = cnz || cur[j] != 0.0f;
      }
      cur_zero = __ballot(cnz) == 0ull;
    }
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) out[(long)n * C + lane * CPL + j] = cur[j];
}

}  // namespace bq

using namespace bq;

extern "C" __attribute__((visibility("default"))) int bq_project_points(
    const float *points, const float *depth, const float *frames, int *pix, int F, int N, int W, int H, float fx, float fy,
    float cx, float cy, float depth_min, float depth_max, float accuracy, void *stream) {
  BQ_REQUIRE(points && depth && frames && pix, BQ_EINVAL, "project_points: null pointer");
  BQ_REQUIRE(F > 0 && N > 0 && W > 0 && H > 0 && F <= 65535, BQ_EINVAL, "project_points: bad extents");
  static_assert(sizeof(ProjFrame) == 40 * 4, "ProjFrame is 40 floats (bqhip_fusion.h)");
  hipLaunchKernelGGL(project_points_kernel, dim3((N + 255) / 256, F), dim3(256), 0, (hipStream_t)stream, points, depth,
                     (const ProjFrame *)frames, pix, N, W, H, fx, fy, cx, cy, depth_min, depth_max, accuracy);
  return check_launch("project_points");
}

extern "C" __attribute__((visibility("default"))) int bq_fuse_point_features(
    const int *pix, const float *feat, float *out, int F, int N, int HW, int C, int maxpool, void *stream) {
  BQ_REQUIRE(pix && feat && out, BQ_EINVAL, "fuse_point_features: null pointer");
  BQ_REQUIRE(F > 0 && N > 0 && HW > 0, BQ_EINVAL, "fuse_point_features: bad extents");
  BQ_REQUIRE(C == 64 || C == 128 || C == 256, BQ_EINVAL, "fuse_point_features: %d channels (64, 128 or 256)", C);
  const dim3 grid((N + 15) / 16), block(1024);
  hipStream_t st = (hipStream_t)stream;
  if (C == 64) hipLaunchKernelGGL(fuse_point_features_kernel<1>, grid, block, 0, st, pix, feat, out, F, N, HW, maxpool);
  else if (C == 128) hipLaunchKernelGGL(fuse_point_features_kernel<2>, grid, block, 0, st, pix, feat, out, F, N, HW, maxpool);
  else hipLaunchKernelGGL(fuse_point_features_kernel<4>, grid, block, 0, st, pix, feat, out, F, N, HW, maxpool);
  return check_launch("fuse_point_features");
}
