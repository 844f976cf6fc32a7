cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c36
{
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_bench_gpu.py tests/test_configs_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -8
run() { tag="$1"; dir="$2"; shift; shift; (cd $dir; BQ_PIPE_TRACE=1 python "$@" 2>$GRAFT_REPO_ROOT/gpurun_out/c36/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' $GRAFT_REPO_ROOT/gpurun_out/c36/t.err | sed 's/.*det_fwd/det_fwd/' | cut -c1-300)"); }
for i in 1 2 3; do
  run r03 _r03 bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run HEAD . bench.py --steps 30 --warmup 5 --no-cpu-baseline
done
} > gpurun_out/c36/log.txt 2>&1
cat gpurun_out/c36/log.txt
