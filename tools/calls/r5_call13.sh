cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c13; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "inverted or grid_ball" 2>&1 | tail -5
timeout 600 python tools/dbg_determinism.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tail -6 | cut -c1-300
timeout 900 python bench.py --steps 30 --warmup 5 --no-loop-reference 2>$O/bench.err | tail -1 | cut -c1-600
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_configs_gpu.py::test_c5_per_rank_phased_step_and_peak_hbm 2>&1 | tail -8
echo "== c5 (last: a fault wedges the box)"
timeout 600 python tools/dbg_c5.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | tail -5 | cut -c1-300
