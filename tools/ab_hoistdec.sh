cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hoistdec
timeout 900 python -m pytest tests/test_fusion_gpu.py tests/test_generate_gpu.py tests/test_itm_gpu.py tests/test_two_segment_gpu.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2 3; do
for h in 1 0; do
echo "hoist=$h: $(BQ_HOIST_CROSS_KV=$h timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | cut -c1-110)"
done; done
