// How long does one DEPENDENT kernel of a replayed HIP graph take on this box, as a function of its grid and LDS size?
// (DESIGN: the text side of the fusion is a chain of ~800 launches at 5-8 us each.)  hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(float *p, int n) {
  extern __shared__ float sm[];
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (n > 0) sm[threadIdx.x] = p[i % n];
  if (n > 0 && sm[threadIdx.x] == 12345.f) p[i % n] = 1.f;
}
__global__ void k_rw(const float4 *__restrict__ a, float4 *__restrict__ b, int n4) {   // streams n4 float4: a -> b
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    float4 v = a[i]; v.x += 1.f; b[i] = v;
  }
}

static float time_chain(hipStream_t st, int nk, int grid, int block, int lds, float *buf, int n, bool rw, float4 *a, float4 *b, int n4) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < nk; ++i) {
    if (rw) hipLaunchKernelGGL(k_rw, dim3(grid), dim3(block), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n4);
    else hipLaunchKernelGGL(k_touch, dim3(grid), dim3(block), lds, st, buf, n);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 12; ++r) {
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ts[ts.size() / 2] * 1000.f / nk;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float *buf; CK(hipMalloc(&buf, 1 << 24));
  float4 *a, *b; CK(hipMalloc(&a, 64 << 20)); CK(hipMalloc(&b, 64 << 20));
  CK(hipFuncSetAttribute((const void *)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int nk = 400;
  struct { int grid, block, lds; } cfg[] = {{1, 64, 0}, {240, 256, 0}, {240, 256, 72 * 1024}, {240, 256, 144 * 1024}, {960, 256, 0},
                                            {960, 256, 72 * 1024}, {2048, 256, 0}, {2048, 256, 36 * 1024}, {8192, 256, 0}};
  for (auto c : cfg)
    printf("touch grid %5d block %3d lds %6d : %.2f us per dependent launch (no memory)\n", c.grid, c.block, c.lds,
           time_chain(st, nk, c.grid, c.block, c.lds, buf, 0, false, a, b, 0));
  for (auto c : cfg)
    printf("touch grid %5d block %3d lds %6d : %.2f us per dependent launch (one load per thread)\n", c.grid, c.block, c.lds,
           time_chain(st, nk, c.grid, c.block, c.lds, buf, 1 << 20, false, a, b, 0));
  int sizes[] = {1 << 10, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22};   // float4 counts: 16 KB .. 64 MB
  for (int n4 : sizes)
    printf("stream %8d B through %4d x 256 threads: %.2f us per dependent launch\n", n4 * 16, std::min(2048, (n4 + 255) / 256),
           time_chain(st, nk, std::min(2048, (n4 + 255) / 256), 256, 0, buf, 0, true, a, b, n4));
  return 0;
}
