cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
timeout 1500 python -m pytest tests/test_fusion_gpu.py tests/test_two_segment_gpu.py -q -m gpu 2>&1 | tail -25 > gpurun_out/c2/tests1.log
cat gpurun_out/c2/tests1.log
