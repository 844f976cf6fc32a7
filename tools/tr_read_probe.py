"""Semantics check of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) on gfx950: lane i of a 16-lane group
must receive column i of the 4-row x 16-column block whose row q / columns 4p..4p+3 address is supplied by lane 4q+p."""
import os, subprocess, sys, tempfile, ctypes
import torch
src = r'''
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void k(const __bf16* in, __bf16* out) {
  __shared__ __bf16 tile[64][64];
  for (int i = threadIdx.x; i < 64*64; i += 64) tile[i>>6][i&63] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  __bf16* addr = MODE ? &tile[lane][0] : &tile[q][16 * g + 4 * p];
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)addr);
  *reinterpret_cast<bf16x4*>(out + lane * 4) = v;
}
extern "C" void launch(const void* in, void* out, void* stream) {
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const __bf16*)in, (__bf16*)out);
}
extern "C" void launch1(const void* in, void* out, void* stream) {
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const __bf16*)in, (__bf16*)out);
}
'''
d = tempfile.mkdtemp()
open(os.path.join(d, "t.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "t.hip"), "-o", os.path.join(d, "t.so")])
lib = ctypes.CDLL(os.path.join(d, "t.so"))
r = torch.arange(64).view(64, 1); c = torch.arange(64).view(1, 64)
tile = ((r % 4) * 64 + c).float().to(torch.bfloat16).cuda()
out = torch.zeros(64, 4, dtype=torch.bfloat16, device="cuda")
lib.launch(ctypes.c_void_p(tile.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
lane = torch.arange(64).view(64, 1); j = torch.arange(4).view(1, 4)
want = (j * 64 + 16 * (lane // 16) + lane % 16).float()
print("matches the documented map:", torch.equal(out.float().cpu(), want))
for name, t in (("rows", r.expand(64, 64)), ("cols", c.expand(64, 64))):
    tile = t.float().to(torch.bfloat16).contiguous().cuda()
    lib.launch(ctypes.c_void_p(tile.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print(name, out.float().cpu()[:20].t().tolist())

print("MODE 1: lane L supplies &tile[L][0]; which (lane, element) does each output come from?")
for name, t in (("src lane", r.expand(64, 64)), ("src elem", c.expand(64, 64))):
    tile = t.float().to(torch.bfloat16).contiguous().cuda()
    lib.launch1(ctypes.c_void_p(tile.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print(name, out.float().cpu()[:20].tolist())
