cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c30
timeout 1500 python -m pytest tests/test_fusion_gpu.py tests/test_gemm_gpu.py tests/test_graphed_gpu.py tests/test_pipeline_gpu.py -x -q --tb=short -k "not golden" 2>&1 | grep -v "^  warn\|Warning" | tail -60 > gpurun_out/c30/log.txt
cat gpurun_out/c30/log.txt
