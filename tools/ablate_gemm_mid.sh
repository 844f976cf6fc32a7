#!/bin/bash
# Where the time of the 256 x 128 kernel's weight-gradient form goes: rebuilds csrc/gemm_mid.hip with BQ_MID_ABLATE = 0
# (the kernel), 1 (one MFMA in eight: the memory stream alone) and 2 (no LDS-DMAs: matrix pipe + LDS reads alone) and times
# the per-shape launches.  Run on the GPU box (the library is rebuilt in place; the last build is the plain one).
#   bash tools/ablate_gemm_mid.sh OUTDIR
set -u
out=${1:-gpurun_out/ablate}
mkdir -p "$out"
root=$(cd "$(dirname "$0")/.." && pwd)
objs=$(ls "$root"/bridgeqa_amd/build/*.hip.o | grep -v gemm_mid)
for n in 1 2 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DBQ_MID_ABLATE=$n \
    -I "$root/include" -I "$root/bridgeqa_amd/csrc" -c "$root/bridgeqa_amd/csrc/gemm_mid.hip" -o "$root/bridgeqa_amd/build/gemm_mid.hip.o" || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/bridgeqa_amd/lib/libbqhip.so" $objs "$root/bridgeqa_amd/build/gemm_mid.hip.o" || exit 1
  echo "== BQ_MID_ABLATE=$n" | tee -a "$out/ablate.log"
  timeout 300 python "$root/tools/bench_gemm_dw.py" 5 --quick 2>&1 | grep -v amdgpu.ids | tee -a "$out/ablate.log"
  echo "-- every block's dY aliased (operands MALL-resident)" | tee -a "$out/ablate.log"
  timeout 300 python "$root/tools/bench_gemm_dw.py" 5 --quick --alias 2>&1 | grep -v amdgpu.ids | tee -a "$out/ablate.log"
done
