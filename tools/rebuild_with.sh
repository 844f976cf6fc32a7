#!/bin/bash
# Measurement builds on the GPU box: recompile ONE source of csrc/ with extra -D flags and relink libbqhip.so in place.
#   bash tools/rebuild_with.sh attn "-DBQ_ATTN_DQ_MINW=4"        (no flags: the plain build of that file)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden "$@" \
  -I "$root/include" -I "$root/bridgeqa_amd/csrc" -c "$root/bridgeqa_amd/csrc/$name.hip" -o "$root/bridgeqa_amd/build/$name.hip.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/bridgeqa_amd/lib/libbqhip.so" "$root"/bridgeqa_amd/build/*.hip.o
