cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "fusion_backward_cut or graph_replay_trains" 2>&1 | tail -12 > gpurun_out/c4/tests.log
cat gpurun_out/c4/tests.log
for cut in -1 6 4 8 -1 6; do
  BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --fusion-cut $cut 2>gpurun_out/c4/cut$cut.err | cut -c1-150 > gpurun_out/c4/cut$cut.json
  echo "cut $cut: $(cut -c60-150 gpurun_out/c4/cut$cut.json)"; grep "GPU ms" gpurun_out/c4/cut$cut.err | cut -c60-
done
