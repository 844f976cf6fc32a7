cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c23
timeout 600 python -X faulthandler -m pytest tests/test_graphed_gpu.py -x -q -m gpu -k module_api > gpurun_out/c23/log.txt 2>&1
grep -E "passed|failed|Fatal|Error" gpurun_out/c23/log.txt | tail -5
