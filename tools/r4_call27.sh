cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c27
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "transposed or transpose or dx" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 900 python -m pytest tests/test_fusion_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_graphed_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c27/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c27/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-230)"; }
for i in 1 2; do
  run T-off tools/ab_bench.py fusion_state.TRANSPOSED_DX[0]=False -- --steps 30 --warmup 5 --no-cpu-baseline
  run HEAD bench.py --steps 30 --warmup 5 --no-cpu-baseline
done
} > gpurun_out/c27/log.txt 2>&1
cat gpurun_out/c27/log.txt
