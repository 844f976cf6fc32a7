cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c35
{
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c35/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c35/t.err | sed 's/.*det_loss/det_loss/' | cut -c1-260)"; }
for i in 1 2; do
  run HEAD bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run dw128 tools/ab_bench.py fusion_wgrad._DW_TILE[0]=128 -- --steps 30 --warmup 5 --no-cpu-baseline
done
echo "== ViT dW 128 vs 256"; timeout 300 python tools/bench_gemm_dw.py 2>&1 | grep -v Warn | tail -16
} > gpurun_out/c35/log.txt 2>&1
cat gpurun_out/c35/log.txt
