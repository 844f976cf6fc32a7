cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c40
{
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_bench_gpu.py tests/test_graphed_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -12
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c40/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c40/t.err | sed 's/.*geometry/geometry/' | cut -c1-300)"; }
for i in 1 2 3; do
  run flush-main tools/ab_bench.py pipeline._FLUSH_ON_DET[0]=False -- --steps 30 --warmup 5 --no-cpu-baseline
  run HEAD bench.py --steps 30 --warmup 5 --no-cpu-baseline
done
} > gpurun_out/c40/log.txt 2>&1
cat gpurun_out/c40/log.txt
