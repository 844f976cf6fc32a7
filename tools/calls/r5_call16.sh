cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c16; mkdir -p $O
timeout 900 python -m pytest tests/test_modules_gpu.py -q -x -k "fused_sharedmlp or native_sharedmlp" 2>&1 | tail -15
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/c2.err | cut -c1-220
timeout 900 python bench.py --steps 30 --warmup 5 --no-loop-reference --no-cpu-baseline 2>$O/c3.err | cut -c1-220
