cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
timeout 1500 python -m pytest tests/test_graphed_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do
BQ_PIPE_TRACE=1 python bench.py --loop reference --steps 20 --warmup 5 2>gpurun_out/c21/ref.err | cut -c62-150; grep "GPU ms" gpurun_out/c21/ref.err | cut -c60-
done
