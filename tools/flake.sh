cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flake
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -x -q -m gpu --tb=long 2>&1 | grep -v "Warning\|^  warn" > gpurun_out/flake/run$i.log
  grep -h "passed\|failed\|Aborted" gpurun_out/flake/run$i.log | tail -2
  if grep -q "failed" gpurun_out/flake/run$i.log; then tail -150 gpurun_out/flake/run$i.log | cut -c1-300 > gpurun_out/flake/failure.txt; break; fi
done
