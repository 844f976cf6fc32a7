cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c33
{
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c33/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c33/t.err | sed 's/.*det_loss/det_loss/' | cut -c1-260)"; }
for i in 1 2; do
  bash tools/rebuild_with.sh transpose -DBQ_TRANSPOSE_NT=0
  run plain bench.py --steps 30 --warmup 5 --no-cpu-baseline
  bash tools/rebuild_with.sh transpose
  run nt bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run T-off tools/ab_bench.py fusion_state.TRANSPOSED_DX[0]=False -- --steps 30 --warmup 5 --no-cpu-baseline
  run vitT tools/ab_bench.py fusion_ops._DX_T_ROWS[0]=1000000 -- --steps 30 --warmup 5 --no-cpu-baseline
done
timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q -k "transpose" 2>&1 | grep -E "passed|failed" | tail -2
} > gpurun_out/c33/log.txt 2>&1
cat gpurun_out/c33/log.txt
