cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c34
{
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_fusion_gpu.py tests/test_pipeline_gpu.py tests/test_graphed_gpu.py tests/test_two_segment_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -12
run() { tag="$1"; dir="$2"; shift; shift; (cd $dir; BQ_PIPE_TRACE=1 python "$@" 2>$GRAFT_REPO_ROOT/gpurun_out/c34/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' $GRAFT_REPO_ROOT/gpurun_out/c34/t.err | sed 's/.*det_loss/det_loss/' | cut -c1-260)"); }
for i in 1 2; do
  run r03 _r03 bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run HEAD . bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run textT . tools/ab_bench.py fusion_ops._DX_T_ROWS[0]=1024 -- --steps 30 --warmup 5 --no-cpu-baseline
done
run ref . bench.py --loop reference --steps 20 --warmup 5 --no-cpu-baseline
run c5 . bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline
} > gpurun_out/c34/log.txt 2>&1
cat gpurun_out/c34/log.txt
