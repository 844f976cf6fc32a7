// Multi-tensor bf16 transpose: the K-contiguous operand of the text side's input-gradient GEMMs.
//
// dX = dY W (reference models/med.py:112-118,226-232,292-317 -- the linears of BertSelfAttention / BertSelfOutput /
// BertIntermediate / BertOutput under autograd: grad_input = grad_output.mm(weight)) contracts over W's ROWS.  Reading the (N, K) operand "down" costs the small-M
// launches of the fusion backward twice the time of their forward twins (tools/bench_small_gemm.py: 25.2 vs 13.8 us at a
// 3072-row contraction, 15.1 vs 10.9 at 2304; every K tile touches a new 98 KB region, 64-B pieces per row), so the text
// side keeps a second bf16 copy W^T (K, N) of each such weight and runs dX on the forward's K-contiguous form.  This
// kernel refreshes ALL of them in one launch per optimizer step, off the critical path (the text_prep phase of
// pipeline.PhasedTrainStep, beside the image encoder): 64 x 64 tiles through LDS, 128-B rows in, 128-B rows out.
#include "bq_common.h"

// measurement builds only (tools/rebuild_with.sh transpose -DBQ_TRANSPOSE_NT=0): plain loads / stores
#ifndef BQ_TRANSPOSE_NT
#define BQ_TRANSPOSE_NT 1
#endif

namespace bq {

struct TransposeTensor {
  const __bf16 *src;   // (N, K), row stride ld
  __bf16 *dst;         // (K, N) contiguous
  int N, K, ld, tiles_k;
};

constexpr int TR_PAD = 66;   // LDS row stride in elements: 33 dwords, column reads of 2-byte elements spread over the banks

__global__ __launch_bounds__(256) void transpose_multi_bf16_kernel(const TransposeTensor *__restrict__ table,
                                                                   const int2 *__restrict__ chunks, int n_chunks) {
  __shared__ __bf16 tile[64 * TR_PAD];
  const int r = threadIdx.x >> 2, seg = (threadIdx.x & 3) * 16;   // row of the tile, first of this lane's 16 elements
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // (a grid smaller than the tile count walks the tiles: the "gentle" launch beside a latency-bound kernel chain)
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
  const int2 ck = chunks[c];
  const TransposeTensor T = table[ck.x];
  const int n0 = (ck.y / T.tiles_k) * 64, k0 = (ck.y % T.tiles_k) * 64;
  {
    const __bf16 *s = T.src + (long)(n0 + r) * T.ld + k0 + seg;
    // (nontemporal: 1.1 GB streamed once per step beside a kernel chain that lives on its cache-resident operands)
#if BQ_TRANSPOSE_NT
    const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s));
    const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s + 8));
#else
    const u32x4 a = *reinterpret_cast<const u32x4 *>(s), b = *reinterpret_cast<const u32x4 *>(s + 8);
#endif
    unsigned *d = reinterpret_cast<unsigned *>(tile + r * TR_PAD + seg);   // (4-byte aligned: TR_PAD and seg are even)
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = a[e]; d[4 + e] = b[e]; }
  }
  __syncthreads();
  {
    // output row k0 + r, columns n0 + seg .. + 16: column r of the tile, rows seg .. seg + 16
    unsigned short v[16];
    const unsigned short *t16 = reinterpret_cast<const unsigned short *>(tile);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = t16[(seg + e) * TR_PAD + r];
    u32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (unsigned)v[2 * e] | ((unsigned)v[2 * e + 1] << 16);
      b[e] = (unsigned)v[8 + 2 * e] | ((unsigned)v[8 + 2 * e + 1] << 16);
    }
    __bf16 *o = T.dst + (long)(k0 + r) * T.N + n0 + seg;
#if BQ_TRANSPOSE_NT
    __builtin_nontemporal_store(a, reinterpret_cast<u32x4 *>(o));
    __builtin_nontemporal_store(b, reinterpret_cast<u32x4 *>(o + 8));
#else
    *reinterpret_cast<u32x4 *>(o) = a;
    *reinterpret_cast<u32x4 *>(o + 8) = b;
#endif
  }
  __syncthreads();   // (the next tile overwrites the image)
  }
}

}  // namespace bq
using namespace bq;

extern "C" __attribute__((visibility("default"))) int bq_transpose_tensor_bytes(void) { return (int)sizeof(TransposeTensor); }

// table: n_tensors records {src, dst (pointers), N, K, ld, tiles_k = K / 64 (int32)} in DEVICE memory (layout = struct
// TransposeTensor, bq_transpose_tensor_bytes() each; N and K multiples of 64, ld a multiple of 8, 16-byte aligned
// pointers: the CALLER guarantees it, the table is not read on the host); chunks: n_chunks x {tensor index, tile index}
// int32 pairs covering every 64 x 64 tile (tile index = (n / 64) * tiles_k + k / 64).  max_wgs > 0: at most that many
// workgroups, each walking several tiles (<= 0: one workgroup per tile).
extern "C" __attribute__((visibility("default"))) int bq_transpose_multi_bf16(const void *table, const void *chunks, int n_chunks,
                                                                              int max_wgs, void *stream) {
  BQ_REQUIRE(n_chunks >= 0, BQ_EINVAL, "transpose_multi: bad chunk count");
  if (n_chunks == 0) return BQ_OK;
  BQ_REQUIRE(table && chunks, BQ_EINVAL, "transpose_multi: null pointer");
  const int grid = (max_wgs > 0 && max_wgs < n_chunks) ? max_wgs : n_chunks;
  hipLaunchKernelGGL(transpose_multi_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const TransposeTensor *)table, (const int2 *)chunks, n_chunks);
  return check_launch("transpose_multi");
}
