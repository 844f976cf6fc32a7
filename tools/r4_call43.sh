cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c43
timeout 900 python -m pytest tests/test_fusion_gpu.py tests/test_pipeline_gpu.py tests/test_gemm_gpu.py -x -q --tb=short -k "transpos or graph_replay or twin or hoisted" 2>&1 | grep -v "Warning\|^  warn" | tail -6 > gpurun_out/c43/log.txt
cat gpurun_out/c43/log.txt
