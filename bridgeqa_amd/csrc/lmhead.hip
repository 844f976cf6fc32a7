// LM head + label-smoothed cross entropy without an fp32 logits tensor (reference models/med.py:1417-1432:
// prediction_scores = cls(sequence_output); shifted; CrossEntropyLoss(reduction='none', label_smoothing=0.1); and
// models/med.py:663-707 BertLMPredictionHead: Linear(768, 30524) tied to the word embeddings + bias).
//
// forward : bq_gemm_bf16 with BQ_GEMM_EPI_BIAS_CE (csrc/gemm.hip) computes the logits tile by tile on the MFMA pipeline,
//           stores them ONCE as bf16 (rows x padded vocabulary, 9.8 MB at config c3) and leaves, from the fp32
//           accumulators, per (vocabulary half tile, row) the triple (max, sum exp(z - max), sum z) plus the target logit;
//           lmhead_ce_combine_kernel merges the 240 triples of a row into lse and the loss.
// backward: lmhead_ce_dlogits_kernel turns the stored logits IN PLACE into dlogits = g * (softmax - (1-eps) onehot -
//           eps / V); the weight / hidden / bias gradients are then three launches of the GEMM family
//           (dW = dlogits^T h, dH = dlogits W with the 30 528-long contraction split over workgroups, db = column sums).
#include "bq_common.h"
#include "bqhip_fusion.h"

namespace bq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one thread per row: merge nrec (max, sumexp, sumz) records
// one WAVE per row: lane g folds records g, g+64, ... (independent loads in flight instead of one serial chain of nrec
// dependent HBM round trips per thread: 81 us -> a few us at nrec = 240), then the 64 (max, sum) pairs are merged by
// xor-shuffles -- the merge of two online-softmax states is associative and commutative up to rounding, and the shuffle
// tree is fixed, so the result is deterministic
__global__ __launch_bounds__(256) void lmhead_ce_combine_kernel(const float *__restrict__ part, const float *__restrict__ zt,
                                                                const int *__restrict__ tgt, float *__restrict__ loss,
                                                                float *__restrict__ lse, int R, int nrec, int V,
                                                                float smoothing) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;  // wave-uniform
  float m = -INFINITY, s = 0.f, sz = 0.f;
  for (int g = lane; g < nrec; g += 64) {
    const float *p = part + ((long)g * R + r) * 3;
    const float mg = p[0], sg = p[1];
    sz += p[2];
    if (mg == -INFINITY) continue;  // a half tile with no valid vocabulary entry
    const float mn = fmaxf(m, mg);
    s = s * (m == -INFINITY ? 0.f : __expf(m - mn)) + sg * __expf(mg - mn);
    m = mn;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float mo = __shfl_xor(m, off), so = __shfl_xor(s, off);
    sz += __shfl_xor(sz, off);
    const float mn = fmaxf(m, mo);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (mo == -INFINITY ? 0.f : so * __expf(mo - mn));
    m = mn;
  }
  if (lane != 0) return;
  const float l = m + __logf(s);
  lse[r] = l;
  const int t = tgt[r];
  loss[r] = t >= 0 ? (1.0f - smoothing) * (l - zt[r]) + smoothing * (l - sz / (float)V) : 0.f;
}

// logits (R, ld) bf16 -> dlogits in place: g[r] * (exp(z - lse) - (1 - eps) [v == t] - eps / V) for v < V and a valid
// target, 0 elsewhere (ignored rows, padding columns).  Thread = 8 adjacent vocabulary entries of one row.
__global__ __launch_bounds__(256) void lmhead_ce_dlogits_kernel(__bf16 *__restrict__ z, const float *__restrict__ lse,
                                                                const int *__restrict__ tgt, const float *__restrict__ g,
                                                                int R, int V, int ld, float smoothing) {
  const int cpr = ld >> 3;  // 16-byte chunks per row
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)R * cpr) return;
  const int r = (int)(idx / cpr), v0 = (int)(idx % cpr) * 8;
  bf16x8 *p = reinterpret_cast<bf16x8 *>(z + (long)r * ld + v0);
  const int t = tgt[r];
  bf16x8 out;
  if (t < 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (__bf16)0.f;
  } else {
    const bf16x8 x = *p;
    const float l = lse[r], gr = g[r], un = smoothing / (float)V;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int v = v0 + e;
      float d = __expf((float)x[e] - l) - un - (v == t ? 1.0f - smoothing : 0.f);
      out[e] = (__bf16)(v < V ? gr * d : 0.f);
    }
  }
  *p = out;
}

}  // namespace bq

extern "C" int bq_lmhead_ce_combine(const float *partial, const float *target_logit, const int *target, float *loss,
                                    float *lse, int R, int nrec, int V, float label_smoothing, void *stream) {
  using namespace bq;
  BQ_REQUIRE(partial && target_logit && target && loss && lse && R > 0 && nrec > 0 && V > 0, BQ_EINVAL,
             "lmhead_ce_combine: bad arguments");
  hipLaunchKernelGGL(lmhead_ce_combine_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial,
                     target_logit, target, loss, lse, R, nrec, V, label_smoothing);
  return check_launch("lmhead_ce_combine");
}

extern "C" int bq_lmhead_ce_dlogits(void *logits, const float *lse, const int *target, const float *grad_loss, int R,
                                    int V, int ld, float label_smoothing, void *stream) {
  using namespace bq;
  BQ_REQUIRE(logits && lse && target && grad_loss && R > 0 && V > 0 && ld >= V && ld % 8 == 0 &&
                 ((uintptr_t)logits % 16 == 0), BQ_EINVAL, "lmhead_ce_dlogits: bad arguments");
  const long threads = (long)R * (ld / 8);
  hipLaunchKernelGGL(lmhead_ce_dlogits_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (__bf16 *)logits, lse, target, grad_loss, R, V, ld, label_smoothing);
  return check_launch("lmhead_ce_dlogits");
}
