cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c42
{
timeout 1500 python -m pytest tests/test_fusion_gpu.py tests/test_pipeline_gpu.py tests/test_graphed_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -8
BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/c42/t.err | cut -c62-105; grep 'GPU ms' gpurun_out/c42/t.err | sed 's/.*geometry/geometry/' | cut -c1-300
} > gpurun_out/c42/log.txt 2>&1
cat gpurun_out/c42/log.txt
