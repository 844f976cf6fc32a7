// Multi-tensor AdamW for gfx950 that also writes the bf16 operand copies ("shadows") of the updated parameters.
//
// The optimizer step of the hot path (reference scripts/train.py:410-417: torch.optim.AdamW, betas (0.9, 0.999),
// eps 1e-8, decoupled weight decay) is pure streaming: read p, g, m, v (16 B / parameter), write p, m, v (12 B).  The
// bf16 GEMM path needs a bf16 copy of every updated weight for the next forward; emitting it here costs 2 B / parameter
// instead of a second pass that re-reads p (6 B / parameter, bridgeqa_amd/fusion_ops.refresh_shadows).
// One launch for all tensors: a chunk table maps workgroups to (tensor, offset); 4 x fp32 per lane per iteration.
//
// Arithmetic = torch's (single-tensor definition, torch/optim/adamw.py, amsgrad = False, maximize = False):
//   p  <- p * (1 - lr * wd)
//   m  <- beta1 * m + (1 - beta1) * g            v <- beta2 * v + (1 - beta2) * g * g
//   p  <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with t read from device memory (graph-replay safe: the caller increments it on the stream before this launch).
#include <cstdint>
#include <cstdlib>
#include "bq_common.h"

namespace bq {

struct AdamWTensor {
  float *p;
  const float *g;
  float *m;
  float *v;
  __bf16 *shadow;  // may be null
  long n;
  float lr, wd;
};

#ifndef BQ_ADAMW_CHUNK
#define BQ_ADAMW_CHUNK 8192   // (measurement builds: tools/rebuild_with.sh adamw -DBQ_ADAMW_CHUNK=...)
#endif
constexpr int ADAMW_CHUNK = BQ_ADAMW_CHUNK;  // elements per workgroup

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// NT: every stream of this kernel is touched exactly once per step -- nontemporal loads / stores keep the 10 GB it
// moves from evicting the L2 / MALL lines of whatever runs next (default; BQ_ADAMW_NT=0 selects plain accesses)
template <bool NT>
__device__ __forceinline__ float4 ld4(const float *p) {
  if (NT) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
  }
  return *reinterpret_cast<const float4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, float4 x) {
  if (NT) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
  } else {
    *reinterpret_cast<float4 *>(p) = x;
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(const AdamWTensor *__restrict__ table,
                                                    const int2 *__restrict__ chunks, const float *__restrict__ step,
                                                    float beta1, float beta2, float eps, float clip) {
  const int2 ck = chunks[blockIdx.x];
  const AdamWTensor T = table[ck.x];
  const long off = (long)ck.y * ADAMW_CHUNK;
  const long end = min(T.n, off + ADAMW_CHUNK);
  const float t = step[0];
  const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
  const float step_size = T.lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - T.lr * T.wd;
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  const bool vec = ((T.n & 3) == 0) && ((((uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v) & 15) == 0) &&
                   (!T.shadow || (((uintptr_t)T.shadow & 7) == 0));
  if (vec) {
    for (long i = off + threadIdx.x * 4; i < end; i += 256 * 4) {
      float4 p = ld4<NT>(T.p + i);
      const float4 g = ld4<NT>(T.g + i);
      float4 m = ld4<NT>(T.m + i), v = ld4<NT>(T.v + i);
      float *pp = &p.x, *mm = &m.x, *vv = &v.x;
      const float *gg = &g.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float q = pp[j] * decay;
        const float gj = fminf(fmaxf(gg[j], -clip), clip);  // clip_grad_value_ (lib/solver.py:407-409); clip = inf: off
        mm[j] = beta1 * mm[j] + omb1 * gj;
        vv[j] = beta2 * vv[j] + omb2 * gj * gj;
        const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
        pp[j] = q - step_size * (mm[j] / denom);
      }
      st4<NT>(T.p + i, p);
      st4<NT>(T.m + i, m);
      st4<NT>(T.v + i, v);
      if (T.shadow) {
        bf16x4 s;
        s[0] = (__bf16)p.x; s[1] = (__bf16)p.y; s[2] = (__bf16)p.z; s[3] = (__bf16)p.w;
        *reinterpret_cast<bf16x4 *>(T.shadow + i) = s;
      }
    }
  } else {
    for (long i = off + threadIdx.x; i < end; i += 256) {
      const float g = fminf(fmaxf(T.g[i], -clip), clip);
      const float q = T.p[i] * decay;
      const float m = beta1 * T.m[i] + omb1 * g;
      const float v = beta2 * T.v[i] + omb2 * g * g;
      const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
      const float pn = q - step_size * (m / denom);
      T.p[i] = pn; T.m[i] = m; T.v[i] = v;
      if (T.shadow) T.shadow[i] = (__bf16)pn;
    }
  }
}

}  // namespace bq
using namespace bq;

extern "C" __attribute__((visibility("default"))) int bq_adamw_chunk_elems(void) { return ADAMW_CHUNK; }
extern "C" __attribute__((visibility("default"))) int bq_adamw_tensor_bytes(void) { return (int)sizeof(AdamWTensor); }

// table: n_tensors records {p, g, m, v, shadow|NULL (pointers), n (int64), lr, wd (f32)} in DEVICE memory (layout =
// struct AdamWTensor above, bq_adamw_tensor_bytes() each); chunks: n_chunks x {tensor index, chunk index} int32 pairs
// in device memory covering every tensor in pieces of bq_adamw_chunk_elems() elements; step: device f32, the 1-based
// step count t of THIS update.
extern "C" __attribute__((visibility("default"))) int bq_adamw_multi(const void *table, const void *chunks, int n_chunks,
                                                                     const float *step, float beta1, float beta2,
                                                                     float eps, float grad_clip_value, void *stream) {
  BQ_REQUIRE(n_chunks >= 0, BQ_EINVAL, "adamw: bad chunk count");
  if (n_chunks == 0) return BQ_OK;
  BQ_REQUIRE(table && chunks && step, BQ_EINVAL, "adamw: null pointer");
  const float clip = grad_clip_value > 0.f ? grad_clip_value : INFINITY;
  constexpr bool nt = true;  // default on: 1.95 -> 1.83 ms at 354 M parameters (tools/bench_adamw.py)
  if (nt)
    hipLaunchKernelGGL(adamw_kernel<true>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamWTensor *)table, (const int2 *)chunks, step, beta1, beta2, eps, clip);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream,
                       (const AdamWTensor *)table, (const int2 *)chunks, step, beta1, beta2, eps, clip);
  return check_launch("adamw");
}
