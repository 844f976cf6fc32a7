cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "split_image_backward" 2>&1 | grep -E "Error|assert|passed|failed|pipeline.py:|ddp.py:" | cut -c1-500 | head -12
for i in 1 2 3; do
  (cd _r03 && BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>../gpurun_out/c15/t.err | cut -c62-150; echo "   [r03 code]"; grep "GPU ms" ../gpurun_out/c15/t.err | cut -c60-)
  BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/c15/t.err | cut -c62-150; echo "   [HEAD]"; grep "GPU ms" gpurun_out/c15/t.err | cut -c60-
done
