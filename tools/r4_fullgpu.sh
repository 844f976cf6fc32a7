cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | cut -c1-500 > gpurun_out/full/tests.log
cat gpurun_out/full/tests.log
