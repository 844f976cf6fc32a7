cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -70 | cut -c1-400 > gpurun_out/full/tests.log
cat gpurun_out/full/tests.log
