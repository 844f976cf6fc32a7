cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14
timeout 2400 python -m pytest tests/test_pipeline_gpu.py tests/test_fusion_gpu.py tests/test_configs_gpu.py tests/test_graphed_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -8 | cut -c1-600 > gpurun_out/c14/tests.log; cat gpurun_out/c14/tests.log
for i in 1 2; do
for a in "" "--no-text-prologue"; do
  BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $a 2>gpurun_out/c14/t.err | cut -c62-150; echo "   [$a]"; grep "GPU ms" gpurun_out/c14/t.err | cut -c60-
done; done
