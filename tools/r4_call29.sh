cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c29
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "transpose" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -4
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c29/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c29/t.err | sed 's/.*det_loss/det_loss/' | cut -c1-260)"; }
for i in 1 2; do
  run T-off tools/ab_bench.py fusion_state.TRANSPOSED_DX[0]=False -- --steps 30 --warmup 5 --no-cpu-baseline
  run wgs0 tools/ab_bench.py pipeline._T_REFRESH_WGS[0]=0 -- --steps 30 --warmup 5 --no-cpu-baseline
  run wgs256 bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run wgs96 tools/ab_bench.py pipeline._T_REFRESH_WGS[0]=96 -- --steps 30 --warmup 5 --no-cpu-baseline
done
timeout 600 python tools/dump_flush_plan.py > gpurun_out/c29/flush_plan.txt 2>&1
} > gpurun_out/c29/log.txt 2>&1
cat gpurun_out/c29/log.txt
