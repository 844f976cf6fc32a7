/*
 * bqhip.h -- C ABI of libbqhip.so, the MI355X (gfx950) implementation of BridgeQA's native
 * hot-path operators.
 *
 * Each entry point replaces one function of the reference's pybind11 module `pointnet2._ext`
 * (lib/pointnet2/_ext_src/src/bindings.cpp:6-19) and is what a binding for that module would
 * call: raw DEVICE pointers, extents, and the HIP stream to enqueue on.  No torch types.
 *
 * Contract (mirrors SURVEY.md §8b-ii):
 *   - all tensors dense row-major, float32 / int32, on the current device;
 *   - inputs are borrowed and never written; outputs are caller-allocated;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *     nothing here synchronises, allocates, or keeps global state (graph-capture safe);
 *   - return value: 0 on success, a negative BQ_E* code on a bad argument, or a positive
 *     hipError_t if the launch failed.  The reference prints and exit(-1)s on a launch failure
 *     (include/cuda_utils.h:30-39); here the caller must raise.  bq_last_error() returns a
 *     thread-local description of the last non-zero status.
 *   - "zero_init" notes: the reference's C++ wrappers allocate outputs with torch::zeros
 *     (ball_query.cpp:19-21, group_points.cpp:48-50, ...).  Functions below state whether they
 *     require that (grad scatter-adds do; ball query writes every slot itself).
 */
#ifndef BQHIP_H
#define BQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BQ_OK 0
#define BQ_EINVAL (-1)   /* bad extent / null pointer */
#define BQ_ELIMIT (-2)   /* extent beyond what the kernels support (see each function) */

/* 2 (round 4): bq_gemm_desc grew the batched-row maps (q_rpb / q_bstride / o_rpb / o_bstride), bq_ball_query_background,
 * tile 128 and BQ_GEMM_BACKGROUND arrived after 1 without a bump: a stale libbqhip.so must fail the version check, not a
 * symbol lookup or an EINVAL at its first launch.
 * 3 (round 4): bq_pwconv_bn_fwd took `center`, bq_transpose_multi_bf16 / bq_transpose_tensor_bytes arrived. */
#define BQHIP_ABI_VERSION 6

#if defined(__GNUC__)
#define BQ_API __attribute__((visibility("default")))
#else
#define BQ_API
#endif

BQ_API int bq_abi_version(void);
BQ_API const char *bq_last_error(void);

/* cuda_utils.h:15-19  opt_n_threads(work) = clamp(2^trunc(log(work)/log 2), 1, 512).
 * Exposed because the FPS tie order depends on it (see bq_furthest_point_sampling). */
BQ_API int bq_opt_n_threads(int work_size);

/* furthest_point_sampling  (sampling.cpp:66-87, sampling_gpu.cu:69-229)
 *   xyz (B,N,3) f32 -> idx (B,m) i32.  `workspace` is device scratch of at least
 *   bq_fps_workspace_bytes(B,N) bytes (0 for small scenes: pass NULL); it plays the role of the
 *   reference's `temp (B,N)` tensor but holds the spatially sorted scene (contents ignored on entry).
 *   Result is index-exact w.r.t. the reference kernel: idx[:,0]=0; points with
 *   x*x+y*y+z*z <= 1e-3 are never selected; exact distance ties resolve by
 *   (bitrev_{log2 bs}(k mod bs), k) ascending, bs = bq_opt_n_threads(N).  Requires N < 2^22. */
BQ_API size_t bq_fps_workspace_bytes(int B, int N);
BQ_API int bq_furthest_point_sampling(const float *xyz, void *workspace, size_t workspace_bytes,
                               int32_t *idx, int B, int N, int m, void *stream);

/* Same result by the unpruned (every point, every round) kernels; temp (B,N) f32 scratch for
 * N > 24576.  Kept for A/B measurement and as an independent on-GPU cross-check. */
BQ_API int bq_furthest_point_sampling_bruteforce(const float *xyz, float *temp, int32_t *idx, int B,
                                          int N, int m, void *stream);

/* gather_points  (sampling.cpp:15-38, sampling_gpu.cu:8-30)   out[b,c,j] = points[b,c,idx[b,j]]
 *   points (B,C,N) f32, idx (B,M) i32, out (B,C,M) f32 */
BQ_API int bq_gather_points(const float *points, const int32_t *idx, float *out, int B, int C, int N,
                     int M, void *stream);

/* gather_points_grad  (sampling.cpp:40-65, sampling_gpu.cu:34-57)
 *   grad_points[b,c,idx[b,j]] += grad_out[b,c,j];  grad_points (B,C,N) MUST be zero_init. */
BQ_API int bq_gather_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int B,
                          int C, int N, int M, void *stream);

/* ball_query  (ball_query.cpp:8-32, ball_query_gpu.cu:9-54)
 *   new_xyz (B,M,3), xyz (B,N,3) -> idx (B,M,nsample) i32: the first `nsample` point ids k
 *   (ascending) with |new_xyz-xyz|^2 < radius*radius (fp32, strict); short neighbourhoods are
 *   padded with the first hit; empty ones are all 0.  Every slot is written (no zero_init
 *   needed).  NB argument order follows `_ext.ball_query(new_xyz, xyz, radius, nsample)`. */
BQ_API int bq_ball_query(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M,
                  float radius, int nsample, void *stream);
/* The same query (same kernel, same results) on a grid of about ONE workgroup per CU instead of one per 8 centres: for a
 * call that runs AHEAD of time on a second stream (the next batch's neighbourhoods under the current step,
 * bridgeqa_amd/pipeline.py) beside chains of short latency-bound kernels.  The full grid holds every wave slot of the
 * chip for the length of the scan and the other stream's kernels queue behind it; this one takes 4x as long and leaves
 * three quarters of the slots free (measured in the c3 step: 38.1 -> 37.8 ms). */
BQ_API int bq_ball_query_background(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M,
                                    float radius, int nsample, void *stream);

/* The same result -- index for index -- through a uniform grid over the scene (csrc/ball_query_grid.hip: counting sort by cell,
 * one wave per centre over the cells its ball meets, hits ranked into index order; balls with more than 256 hits fall back to
 * the exhaustive scan of that centre): ~400x fewer distance tests than the scan at SA1's size.  For large N (the Python layer
 * uses it from N = 8192); workspace: device scratch of bq_ball_query_grid_workspace_bytes(B, N) bytes, contents irrelevant.
 * radius > 0. */
BQ_API size_t bq_ball_query_grid_workspace_bytes(int B, int N);
/* ABI 6.  How bq_ball_query_grid bins a scene's points: 1 (default) = three short launches for 4096 <= N < 2^19 (partial
 * bounding boxes; the cell of every point as 16 bits; 16 workgroups per scene that each own a chunk of the cells -- histogram,
 * start table and records of their chunk, no scan across workgroups and no global atomics); 0 = round 5's one workgroup per
 * scene (one launch).  Same indices either way (the order of the records inside a cell does not reach the output).  Returns the
 * previous mode; a measurement switch (tools/time_ball_query.py). */
BQ_API int bq_ball_query_grid_build_mode(int multi);
BQ_API int bq_ball_query_grid(const float *new_xyz, const float *xyz, int32_t *idx, int B, int N, int M, float radius,
                              int nsample, void *workspace, size_t workspace_bytes, void *stream);

/* group_points  (group_points.cpp:12-36, group_points_gpu.cu:8-39)
 *   out[b,c,j,k] = points[b,c,idx[b,j,k]];  points (B,C,N), idx (B,M,S), out (B,C,M,S) */
BQ_API int bq_group_points(const float *points, const int32_t *idx, float *out, int B, int C, int N,
                    int M, int S, void *stream);

/* group_points_grad  (group_points.cpp:38-62, group_points_gpu.cu:43-75)
 *   grad_points[b,c,idx[b,j,k]] += grad_out[b,c,j,k];  grad_points (B,C,N) MUST be zero_init. */
BQ_API int bq_group_points_grad(const float *grad_out, const int32_t *idx, float *grad_points, int B,
                         int C, int N, int M, int S, void *stream);

/* three_nn  (interpolate.cpp:14-40, interpolate_gpu.cu:9-68)
 *   unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3) f32 (SQUARED distances, ascending),
 *   idx (B,n,3) i32; strict '<' => lowest index wins ties; m<3 leaves +inf / 0. */
BQ_API int bq_three_nn(const float *unknown, const float *known, float *dist2, int32_t *idx, int B,
                int n, int m, void *stream);

/* three_interpolate  (interpolate.cpp:42-70, interpolate_gpu.cu:72-111)
 *   out[b,c,j] = p[idx0]*w0 + p[idx1]*w1 + p[idx2]*w2 (left to right, no FMA contraction)
 *   points (B,C,m), idx (B,n,3), weight (B,n,3), out (B,C,n) */
BQ_API int bq_three_interpolate(const float *points, const int32_t *idx, const float *weight,
                         float *out, int B, int C, int m, int n, void *stream);

/* three_interpolate_grad  (interpolate.cpp:72-99, interpolate_gpu.cu:116-154)
 *   grad_points[b,c,idx_t] += grad_out[b,c,j]*w_t;  grad_points (B,C,m) MUST be zero_init. */
BQ_API int bq_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                              float *grad_points, int B, int C, int n, int m, void *stream);

/* ---- fused forms (no reference counterpart as a single op; each equals the composition
 *      the reference's Python performs, cited per function) ------------------------------ */

/* ThreeNN.forward (pointnet2_utils.py:140-142): bq_three_nn followed by a correctly rounded sqrt;
 *   dist (B,n,3) f32 = sqrt(dist2). */
BQ_API int bq_three_nn_dist(const float *unknown, const float *known, float *dist, int32_t *idx, int B,
                     int n, int m, void *stream);

/* QueryAndGroup.forward body after ball_query (pointnet2_utils.py:348-359):
 *   out[b, 0:3, j, k] = (xyz[b, idx[b,j,k], :] - new_xyz[b,j,:]) / radius   (if normalize)
 *   out[b, 3:3+C, j, k] = features[b, :, idx[b,j,k]]
 *   xyz (B,N,3), new_xyz (B,M,3), features (B,C,N) or NULL (C=0), idx (B,M,S),
 *   out (B,3+C,M,S).  Bit-identical to group(xyz^T) - centre, then true division by radius. */
BQ_API int bq_group_concat(const float *xyz, const float *new_xyz, const float *features,
                    const int32_t *idx, float *out, int B, int C, int N, int M, int S,
                    float radius, int normalize, void *stream);

/* Backward of bq_group_concat w.r.t. features and (optionally) xyz / new_xyz:
 *   grad_out (B,3+C,M,S);  grad_features (B,C,N) zero_init (or NULL);
 *   grad_xyz (B,N,3) zero_init (or NULL);  grad_new_xyz (B,M,3) zero_init (or NULL). */
BQ_API int bq_group_concat_grad(const float *grad_out, const int32_t *idx, float *grad_features,
                         float *grad_xyz, float *grad_new_xyz, int B, int C, int N, int M, int S,
                         float radius, int normalize, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BQHIP_H */
