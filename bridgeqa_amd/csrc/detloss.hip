// The detection loss of the training step and its gradient as TWO launches -- SURVEY.md §8f rank 1, reference
// lib/loss_helper.py:25-193 (compute_vote_loss, compute_objectness_loss, compute_box_and_sem_cls_loss) over
// utils/nn_distance.py:6-52.  On torch ops (bridgeqa_amd/loss_helper.py, the golden-pinned mirror) those functions and their
// autograd are ~460 launches of 2-6 us on B x 256 proposals: 1.0 ms of the detector stream in front of the fusion and 1.5 ms
// at the head of the detector's backward, which is the phase that ends the step (profiles/r04_c3_phases.txt).
//
// The terms are normalised by batch-wide sums -- sum(mask), npos, sum(box_label_mask) --, so the forward is two launches of
// the same 64 x 256-thread grid (the whole job is ~1 M distance evaluations; ONE workgroup, the first version, took 0.9 ms):
// det_loss_kernel: per seed / per proposal / per ground-truth box the quantities of the reference, the numerators and
// denominators as per-workgroup partial sums, the gradient of every term with respect to the network's outputs written
// UNNORMALISED; det_loss_finish_kernel: every thread adds the partial sums in block order (bitwise reproducible: no
// atomics), block 0 writes the terms, all scale the gradients by 1 / denominator and add the GT -> proposal share of the
// centre gradient in ground-truth order.
// det_loss_bwd_kernel: gradient buffers x the upstream gradient of their term (the terms are returned separately, so the
// caller's weights -- scripts/train.py:97-103 -- and the x10 stay ordinary autograd arithmetic on eight scalars).
//
// Arithmetic: fp32 throughout, formulas in the reference's form (log-softmax as x - max - log(sum exp(x - max)),
// Huber as 0.5 q^2 + delta (|e| - q)); reductions in a different order than torch's => compared at 1e-5 relative
// (tests/test_detloss_gpu.py against loss_helper.py and tests/golden/det_loss.npz).  torch.min's first-index tie rule kept.
#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

struct DetLossArgs {
  // network outputs (fp32, contiguous)
  const float *seed_xyz;        // (B, S, 3)
  const float *vote_xyz;        // (B, S * VF, 3)
  const float *agg_xyz;         // (B, K, 3)  aggregated_vote_xyz
  const float *obj_scores;      // (B, K, 2)
  const float *center;          // (B, K, 3)
  const float *head_scores;     // (B, K, NH)
  const float *head_res;        // (B, K, NH)  heading_residuals_normalized
  const float *size_scores;     // (B, K, NS)
  const float *size_res;        // (B, K, NS, 3)  size_residuals_normalized
  const float *sem_scores;      // (B, K, NC)
  // labels
  const void *seed_inds;        // (B, S) int32 or int64 (seed_inds_i64)
  const float *vote_label;      // (B, N, 3 * GT_VOTE_FACTOR)
  const long long *vote_mask;   // (B, N) int64
  const float *center_label;    // (B, G, >= 3): row stride cl_ld
  const float *box_mask;        // (B, G)
  const long long *head_cls_label, *size_cls_label, *sem_cls_label;   // (B, G) int64
  const float *head_res_label;  // (B, G)
  const float *size_res_label;  // (B, G, 3)
  const float *mean_size;       // (NS, 3)
  // outputs
  float *terms;                 // [16]: vote, objectness, center, heading_cls, heading_reg, size_cls, size_reg, sem_cls,
                                //       pos_ratio, neg_ratio, (spare)
  long long *obj_label;         // (B, K)
  float *obj_mask;              // (B, K)
  long long *assignment;        // (B, K)
  float *g_vote, *g_obj, *g_center, *g_head_scores, *g_head_res, *g_size_scores, *g_size_res, *g_sem;   // like their inputs
  int *scratch;                 // (B, G) int: nearest proposal of every ground-truth centre; then DL_BLOCKS x 16 float partial sums
  int B, S, VF, N, K, G, NH, NS, NC, cl_ld, seed_inds_i64;
  int ld_obj, ld_hs, ld_hr, ld_ss, ld_sr, ld_sem;   // floats per proposal row of the six score tensors AND of their gradient buffers
                                                    // (slices of one (B, K, channels) head output: all = channels)
  float near_thr, far_thr, w_neg, w_pos, head_bin;   // head_bin = pi / NH
};

constexpr int DL_THREADS = 256, DL_BLOCKS = 64, DL_ALL = DL_THREADS * DL_BLOCKS;
constexpr int DL_SUMS = 12;
enum { S_VOTE = 0, S_VMASK, S_OBJ, S_OMASK, S_NPOS, S_C1, S_C2, S_BM, S_HC, S_HR, S_SC, S_SR };   // + S_SEM kept apart
constexpr int S_SEM = 12;

__device__ __forceinline__ float huber1(float e) {   // delta = 1 (nn_distance.py:6-23)
  const float a = fabsf(e), q = fminf(a, 1.0f);
  return 0.5f * q * q + (a - q);
}

// -log_softmax(x)[y] over n logits (stride 1) and the softmax written to p[] (n <= 32); torch: x - max - log(sum(exp(x - max)))
__device__ __forceinline__ float ce_row(const float *x, int n, int y, float *p) {
  float mx = x[0];
  for (int i = 1; i < n; ++i) mx = fmaxf(mx, x[i]);
  float se = 0.f;
  for (int i = 0; i < n; ++i) { p[i] = expf(x[i] - mx); se += p[i]; }
  const float lse = logf(se), inv = 1.0f / se;
  for (int i = 0; i < n; ++i) p[i] *= inv;
  return -((x[y] - mx) - lse);
}

__global__ __launch_bounds__(DL_THREADS) void det_loss_kernel(const DetLossArgs a) {
  __shared__ float s_part[DL_THREADS / 64][DL_SUMS + 1];
  const int tid = threadIdx.x, gtid = blockIdx.x * DL_THREADS + threadIdx.x;
  float sum[DL_SUMS + 1];
#pragma unroll
  for (int i = 0; i <= DL_SUMS; ++i) sum[i] = 0.f;

  // ---- A. vote loss (loss_helper.py:25-70): per seed, min over (predicted vote, GT vote) of the L1 distance ----------
  for (int it = gtid; it < a.B * a.S; it += DL_ALL) {
    const int b = it / a.S;
    const long ind = a.seed_inds_i64 ? (long)((const long long *)a.seed_inds)[it] : (long)((const int *)a.seed_inds)[it];
    const float m = (float)a.vote_mask[(long)b * a.N + ind];
    const float *sx = a.seed_xyz + (long)it * 3;
    const float *vl = a.vote_label + ((long)b * a.N + ind) * 9;
    float best = 0.f;
    int bv = 0, bj = 0;
    // nn_distance(votes (VF), gt (3), l1) -> dist2[j] = min_v, then min over j: first index wins ties at both levels
    for (int j = 0; j < 3; ++j) {
      const float gx = vl[3 * j] + sx[0], gy = vl[3 * j + 1] + sx[1], gz = vl[3 * j + 2] + sx[2];
      for (int v = 0; v < a.VF; ++v) {
        const float *vx = a.vote_xyz + ((long)it * a.VF + v) * 3;
        const float d = (fabsf(vx[0] - gx) + fabsf(vx[1] - gy)) + fabsf(vx[2] - gz);
        if ((j == 0 && v == 0) || d < best) { best = d; bv = v; bj = j; }
      }
    }
    sum[S_VOTE] += best * m;
    sum[S_VMASK] += m;
    for (int v = 0; v < a.VF; ++v) {
      float *g = a.g_vote + ((long)it * a.VF + v) * 3;
      if (v == bv) {
        const float *vx = a.vote_xyz + ((long)it * a.VF + v) * 3;
        const float gx = vl[3 * bj] + sx[0], gy = vl[3 * bj + 1] + sx[1], gz = vl[3 * bj + 2] + sx[2];
        const float ex = vx[0] - gx, ey = vx[1] - gy, ez = vx[2] - gz;   // d|e|/de = sign(e), 0 at 0 (torch.abs)
        g[0] = m * (float)((ex > 0.f) - (ex < 0.f));
        g[1] = m * (float)((ey > 0.f) - (ey < 0.f));
        g[2] = m * (float)((ez > 0.f) - (ez < 0.f));
      } else {
        g[0] = g[1] = g[2] = 0.f;
      }
    }
  }

  // ---- B. per proposal: objectness (:72-115), box terms and semantic class (:118-193) ----------------------------------
  for (int it = gtid; it < a.B * a.K; it += DL_ALL) {
    const int b = it / a.K;
    const float *gt = a.center_label + (long)b * a.G * a.cl_ld;
    const float *q = a.agg_xyz + (long)it * 3, *c = a.center + (long)it * 3;
    float d1 = 0.f, dc = 0.f;
    int i1 = 0, ic = 0;
    for (int g = 0; g < a.G; ++g) {
      const float gx = gt[g * a.cl_ld], gy = gt[g * a.cl_ld + 1], gz = gt[g * a.cl_ld + 2];
      const float ax = q[0] - gx, ay = q[1] - gy, az = q[2] - gz;
      const float e1 = (ax * ax + ay * ay) + az * az;
      if (g == 0 || e1 < d1) { d1 = e1; i1 = g; }
      const float bx = c[0] - gx, by = c[1] - gy, bz = c[2] - gz;
      const float e2 = (bx * bx + by * by) + bz * bz;
      if (g == 0 || e2 < dc) { dc = e2; ic = g; }
    }
    const float eu = sqrtf(d1 + 1e-6f);
    const bool near = eu < a.near_thr;
    const float lab = near ? 1.f : 0.f, msk = (near || eu > a.far_thr) ? 1.f : 0.f;
    a.obj_label[it] = near ? 1 : 0;
    a.obj_mask[it] = msk;
    a.assignment[it] = i1;
    float p[32];
    {   // weighted cross entropy over the two objectness logits
      const float w = near ? a.w_pos : a.w_neg;
      const float l = ce_row(a.obj_scores + (long)it * a.ld_obj, 2, near ? 1 : 0, p);
      sum[S_OBJ] += (w * l) * msk;
      sum[S_OMASK] += msk;
      a.g_obj[(long)it * a.ld_obj + 0] = w * msk * (p[0] - (near ? 0.f : 1.f));
      a.g_obj[(long)it * a.ld_obj + 1] = w * msk * (p[1] - (near ? 1.f : 0.f));
    }
    sum[S_NPOS] += lab;
    // centre, proposal -> nearest GT centre
    sum[S_C1] += dc * lab;
    {
      const float gx = gt[ic * a.cl_ld], gy = gt[ic * a.cl_ld + 1], gz = gt[ic * a.cl_ld + 2];
      a.g_center[(long)it * 3 + 0] = 2.f * (c[0] - gx) * lab;   // (the GT -> proposal share is added in pass C)
      a.g_center[(long)it * 3 + 1] = 2.f * (c[1] - gy) * lab;
      a.g_center[(long)it * 3 + 2] = 2.f * (c[2] - gz) * lab;
    }
    const long lb = (long)b * a.G + i1;   // labels of the assigned GT box (object_assignment = the objectness loss' ind1)
    {   // heading class + residual
      const int y = (int)a.head_cls_label[lb];
      const float l = ce_row(a.head_scores + (long)it * a.ld_hs, a.NH, y, p);
      sum[S_HC] += l * lab;
      for (int i = 0; i < a.NH; ++i) a.g_head_scores[(long)it * a.ld_hs + i] = lab * (p[i] - (i == y ? 1.f : 0.f));
      const float e = a.head_res[(long)it * a.ld_hr + y] - a.head_res_label[lb] / a.head_bin;
      sum[S_HR] += huber1(e) * lab;
      for (int i = 0; i < a.NH; ++i) a.g_head_res[(long)it * a.ld_hr + i] = i == y ? lab * fminf(fmaxf(e, -1.f), 1.f) : 0.f;
    }
    {   // size class + residual (mean over the 3 coordinates of the Huber loss)
      const int y = (int)a.size_cls_label[lb];
      const float l = ce_row(a.size_scores + (long)it * a.ld_ss, a.NS, y, p);
      sum[S_SC] += l * lab;
      for (int i = 0; i < a.NS; ++i) a.g_size_scores[(long)it * a.ld_ss + i] = lab * (p[i] - (i == y ? 1.f : 0.f));
      float *gr = a.g_size_res + (long)it * a.ld_sr;
      for (int i = 0; i < a.NS * 3; ++i) gr[i] = 0.f;
      float hs = 0.f;
      for (int d = 0; d < 3; ++d) {
        const float e = a.size_res[(long)it * a.ld_sr + y * 3 + d] - a.size_res_label[lb * 3 + d] / a.mean_size[y * 3 + d];
        hs += huber1(e);
        gr[y * 3 + d] = lab * fminf(fmaxf(e, -1.f), 1.f) * (1.0f / 3.0f);
      }
      sum[S_SR] += (hs / 3.0f) * lab;
    }
    {   // semantic class
      const int y = (int)a.sem_cls_label[lb];
      const float l = ce_row(a.sem_scores + (long)it * a.ld_sem, a.NC, y, p);
      sum[S_SEM] += l * lab;
      for (int i = 0; i < a.NC; ++i) a.g_sem[(long)it * a.ld_sem + i] = lab * (p[i] - (i == y ? 1.f : 0.f));
    }
  }

  // ---- C. centre, GT -> nearest proposal (dist2 of nn_distance(pred_center, gt_center)) ------------------------------------
  for (int it = gtid; it < a.B * a.G; it += DL_ALL) {
    const int b = it / a.G, g = it - b * a.G;
    const float *gt = a.center_label + ((long)b * a.G + g) * a.cl_ld;
    const float *c = a.center + (long)b * a.K * 3;
    float d2 = 0.f;
    int k2 = 0;
    for (int k = 0; k < a.K; ++k) {
      const float bx = c[k * 3] - gt[0], by = c[k * 3 + 1] - gt[1], bz = c[k * 3 + 2] - gt[2];
      const float e = (bx * bx + by * by) + bz * bz;
      if (k == 0 || e < d2) { d2 = e; k2 = k; }
    }
    const float bm = a.box_mask[it];
    sum[S_C2] += d2 * bm;
    sum[S_BM] += bm;
    a.scratch[it] = k2;
  }

  // ---- per-workgroup partial sums, fixed order: lanes (butterfly), then waves in index order ------------------------------
#pragma unroll
  for (int i = 0; i <= DL_SUMS; ++i) {
    float v = sum[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((tid & 63) == 0) s_part[tid >> 6][i] = v;
  }
  __syncthreads();
  if (tid <= DL_SUMS) {
    float v = 0.f;
    for (int w = 0; w < DL_THREADS / 64; ++w) v += s_part[w][tid];
    ((float *)(a.scratch + a.B * a.G))[blockIdx.x * 16 + tid] = v;
  }
}

__global__ __launch_bounds__(DL_THREADS) void det_loss_finish_kernel(const DetLossArgs a) {
  __shared__ float s_tot[DL_SUMS + 1];
  const int tid = threadIdx.x, gtid = blockIdx.x * DL_THREADS + threadIdx.x;
  if (tid <= DL_SUMS) {   // the same additions in the same order in every workgroup
    const float *part = (const float *)(a.scratch + a.B * a.G);
    float v = 0.f;
    for (int w = 0; w < DL_BLOCKS; ++w) v += part[w * 16 + tid];
    s_tot[tid] = v;
  }
  __syncthreads();
  const float inv_vm = 1.0f / (s_tot[S_VMASK] + 1e-6f), inv_om = 1.0f / (s_tot[S_OMASK] + 1e-6f);
  const float inv_np = 1.0f / (s_tot[S_NPOS] + 1e-6f), inv_bm = 1.0f / (s_tot[S_BM] + 1e-6f);
  if (gtid == 0) {
    const float total = (float)(a.B * a.K);
    a.terms[0] = s_tot[S_VOTE] * inv_vm;
    a.terms[1] = s_tot[S_OBJ] * inv_om;
    a.terms[2] = s_tot[S_C1] * inv_np + s_tot[S_C2] * inv_bm;
    a.terms[3] = s_tot[S_HC] * inv_np;
    a.terms[4] = s_tot[S_HR] * inv_np;
    a.terms[5] = s_tot[S_SC] * inv_np;
    a.terms[6] = s_tot[S_SR] * inv_np;
    a.terms[7] = s_tot[S_SEM] * inv_np;
    a.terms[8] = s_tot[S_NPOS] / total;                             // pos_ratio
    a.terms[9] = s_tot[S_OMASK] / total - s_tot[S_NPOS] / total;    // neg_ratio
  }

  // ---- normalise the gradients ---------------------------------------------------------------------------------------------
  for (int it = gtid; it < a.B * a.S; it += DL_ALL)
    for (int e = 0; e < a.VF * 3; ++e) a.g_vote[(long)it * a.VF * 3 + e] *= inv_vm;
  for (int it = gtid; it < a.B * a.K; it += DL_ALL) {
    const int b = it / a.K, k = it - b * a.K;
    a.g_obj[(long)it * a.ld_obj] *= inv_om;
    a.g_obj[(long)it * a.ld_obj + 1] *= inv_om;
    // centre: own share / npos + the share of every GT box whose nearest proposal this is / sum(box_label_mask), in GT order
    float gx = a.g_center[(long)it * 3] * inv_np, gy = a.g_center[(long)it * 3 + 1] * inv_np, gz = a.g_center[(long)it * 3 + 2] * inv_np;
    const float *c = a.center + (long)it * 3;
    const int *nk = a.scratch + (long)b * a.G;
    for (int g = 0; g < a.G; ++g)
      if (nk[g] == k) {
        const float *gt = a.center_label + ((long)b * a.G + g) * a.cl_ld;
        const float s = 2.f * a.box_mask[(long)b * a.G + g] * inv_bm;
        gx += s * (c[0] - gt[0]); gy += s * (c[1] - gt[1]); gz += s * (c[2] - gt[2]);
      }
    a.g_center[(long)it * 3] = gx; a.g_center[(long)it * 3 + 1] = gy; a.g_center[(long)it * 3 + 2] = gz;
    for (int i = 0; i < a.NH; ++i) { a.g_head_scores[(long)it * a.ld_hs + i] *= inv_np; a.g_head_res[(long)it * a.ld_hr + i] *= inv_np; }
    for (int i = 0; i < a.NS; ++i) a.g_size_scores[(long)it * a.ld_ss + i] *= inv_np;
    for (int i = 0; i < a.NS * 3; ++i) a.g_size_res[(long)it * a.ld_sr + i] *= inv_np;
    for (int i = 0; i < a.NC; ++i) a.g_sem[(long)it * a.ld_sem + i] *= inv_np;
  }
}

// segment i: out[r * ld + c] = g[r * ld + c] * (term >= 0 ? up[term] : 0), r < rows, c < width -- a gradient buffer, or the
// channel range of one term inside the (B K, channels) buffer of a packed head output (term < 0: channels no term reads)
constexpr int DL_MAX_SEG = 16;
struct DetLossBwdArgs {
  const float *g[DL_MAX_SEG];
  float *out[DL_MAX_SEG];
  int rows[DL_MAX_SEG], width[DL_MAX_SEG], ld[DL_MAX_SEG], term[DL_MAX_SEG];
  const float *up;   // [8] upstream gradients of the eight terms
};

__global__ __launch_bounds__(256) void det_loss_bwd_kernel(const DetLossBwdArgs a) {
  const int t = blockIdx.y;
  const float s = a.term[t] >= 0 ? a.up[a.term[t]] : 0.f;
  const int w = a.width[t], n = a.rows[t] * w, ld = a.ld[t];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = i / w, c = i - r * w;
    a.out[t][(long)r * ld + c] = a.term[t] >= 0 ? a.g[t][(long)r * ld + c] * s : 0.f;
  }
}

}  // namespace bq

extern "C" int bq_det_loss_fwd(const bq_det_loss_desc *d, void *stream) {
  using namespace bq;
  BQ_REQUIRE(d != nullptr, BQ_EINVAL, "bq_det_loss_fwd: null descriptor");
  BQ_REQUIRE(d->B > 0 && d->S > 0 && d->VF > 0 && d->K > 0 && d->G > 0 && d->N > 0, BQ_EINVAL, "bq_det_loss_fwd: empty extent");
  BQ_REQUIRE(d->NH >= 1 && d->NH <= 32 && d->NS >= 1 && d->NS <= 32 && d->NC >= 1 && d->NC <= 32, BQ_ELIMIT,
             "bq_det_loss_fwd: at most 32 heading bins / size clusters / classes (NH %d NS %d NC %d)", d->NH, d->NS, d->NC);
  BQ_REQUIRE(d->cl_ld >= 3, BQ_EINVAL, "bq_det_loss_fwd: center_label rows need >= 3 floats");
  DetLossArgs a;
  a.seed_xyz = d->seed_xyz; a.vote_xyz = d->vote_xyz; a.agg_xyz = d->aggregated_vote_xyz; a.obj_scores = d->objectness_scores;
  a.center = d->center; a.head_scores = d->heading_scores; a.head_res = d->heading_residuals_normalized;
  a.size_scores = d->size_scores; a.size_res = d->size_residuals_normalized; a.sem_scores = d->sem_cls_scores;
  a.seed_inds = d->seed_inds; a.vote_label = d->vote_label; a.vote_mask = (const long long *)d->vote_label_mask;
  a.center_label = d->center_label; a.box_mask = d->box_label_mask;
  a.head_cls_label = (const long long *)d->heading_class_label; a.size_cls_label = (const long long *)d->size_class_label;
  a.sem_cls_label = (const long long *)d->sem_cls_label; a.head_res_label = d->heading_residual_label;
  a.size_res_label = d->size_residual_label; a.mean_size = d->mean_size_arr;
  a.terms = d->terms; a.obj_label = (long long *)d->objectness_label; a.obj_mask = d->objectness_mask;
  a.assignment = (long long *)d->object_assignment;
  a.g_vote = d->g_vote_xyz; a.g_obj = d->g_objectness_scores; a.g_center = d->g_center; a.g_head_scores = d->g_heading_scores;
  a.g_head_res = d->g_heading_residuals_normalized; a.g_size_scores = d->g_size_scores;
  a.g_size_res = d->g_size_residuals_normalized; a.g_sem = d->g_sem_cls_scores; a.scratch = d->scratch;
  const void *need[] = {a.seed_xyz, a.vote_xyz, a.agg_xyz, a.obj_scores, a.center, a.head_scores, a.head_res, a.size_scores,
                        a.size_res, a.sem_scores, a.seed_inds, a.vote_label, a.vote_mask, a.center_label, a.box_mask,
                        a.head_cls_label, a.size_cls_label, a.sem_cls_label, a.head_res_label, a.size_res_label, a.mean_size,
                        a.terms, a.obj_label, a.obj_mask, a.assignment, a.g_vote, a.g_obj, a.g_center, a.g_head_scores,
                        a.g_head_res, a.g_size_scores, a.g_size_res, a.g_sem, a.scratch};
  for (const void *p : need) BQ_REQUIRE(p != nullptr, BQ_EINVAL, "bq_det_loss_fwd: null pointer in the descriptor");
  a.B = d->B; a.S = d->S; a.VF = d->VF; a.N = d->N; a.K = d->K; a.G = d->G; a.NH = d->NH; a.NS = d->NS; a.NC = d->NC;
  a.cl_ld = d->cl_ld; a.seed_inds_i64 = d->seed_inds_i64;
  a.ld_obj = d->ld_objectness_scores; a.ld_hs = d->ld_heading_scores; a.ld_hr = d->ld_heading_residuals_normalized;
  a.ld_ss = d->ld_size_scores; a.ld_sr = d->ld_size_residuals_normalized; a.ld_sem = d->ld_sem_cls_scores;
  BQ_REQUIRE(a.ld_obj >= 2 && a.ld_hs >= a.NH && a.ld_hr >= a.NH && a.ld_ss >= a.NS && a.ld_sr >= 3 * a.NS && a.ld_sem >= a.NC,
             BQ_EINVAL, "bq_det_loss_fwd: a row stride is shorter than its row");
  a.near_thr = d->near_threshold; a.far_thr = d->far_threshold; a.w_neg = d->objectness_weight_neg; a.w_pos = d->objectness_weight_pos;
  a.head_bin = 3.14159265358979323846f / (float)d->NH;
  hipLaunchKernelGGL(det_loss_kernel, dim3(DL_BLOCKS), dim3(DL_THREADS), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(det_loss_finish_kernel, dim3(DL_BLOCKS), dim3(DL_THREADS), 0, (hipStream_t)stream, a);
  return check_launch("bq_det_loss_fwd");
}

extern "C" int bq_det_loss_bwd(const bq_det_loss_seg *seg, int n, const float *upstream, void *stream) {
  using namespace bq;
  BQ_REQUIRE(seg && upstream && n >= 1 && n <= DL_MAX_SEG, BQ_EINVAL, "bq_det_loss_bwd: 1..%d segments", DL_MAX_SEG);
  DetLossBwdArgs a;
  int mx = 0;
  for (int i = 0; i < n; ++i) {
    BQ_REQUIRE(seg[i].out && (seg[i].term < 0 || seg[i].g) && seg[i].rows >= 0 && seg[i].width >= 0 && seg[i].ld >= seg[i].width
               && seg[i].term < 8, BQ_EINVAL, "bq_det_loss_bwd: segment %d", i);
    a.g[i] = seg[i].g; a.out[i] = seg[i].out; a.rows[i] = seg[i].rows; a.width[i] = seg[i].width; a.ld[i] = seg[i].ld;
    a.term[i] = seg[i].term;
    const int e = seg[i].rows * seg[i].width;
    mx = e > mx ? e : mx;
  }
  a.up = upstream;
  int gx = (mx + 255) / 256;
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(det_loss_bwd_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("bq_det_loss_bwd");
}
