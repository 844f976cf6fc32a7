cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python tools/ab_bench.py "$@" -- --steps 30 --warmup 5 --no-cpu-baseline $EXTRA 2>gpurun_out/c16/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c16/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-200)"; }
for i in 1 2; do
  (cd _r03 && BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>../gpurun_out/c16/t.err | cut -c62-105; echo "   [r03 code] $(grep 'GPU ms' ../gpurun_out/c16/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-200)")
  EXTRA="" run HEAD
  EXTRA="" run forked "pipeline._SINGLE_STREAM[0]=False"
  EXTRA="--no-text-prologue" run noprologue
  EXTRA="" run cat_kv "fusion_ops._TWIN_KV[0]=False"
  EXTRA="" run dw64 "fusion_wgrad._SHORT_DW_TILE[0]=64"
done
