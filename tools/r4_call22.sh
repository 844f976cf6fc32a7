cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c22
timeout 1500 python -X faulthandler -m pytest tests/test_graphed_gpu.py tests/test_pipeline_gpu.py -x -v -m gpu > gpurun_out/c22/log.txt 2>&1
grep -E "PASSED|FAILED|ERROR|Fatal|File \"|Segmentation|Memory access" gpurun_out/c22/log.txt | cut -c1-200 | tail -40
