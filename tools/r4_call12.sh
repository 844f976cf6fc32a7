cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c12
run() { BQ_PIPE_TRACE=1 python tools/ab_bench.py "$@" -- --steps 30 --warmup 5 --no-cpu-baseline $EXTRA 2>gpurun_out/c12/tmp.err | cut -c62-150; grep "GPU ms" gpurun_out/c12/tmp.err | cut -c60-; }
echo "base"; run
echo "short dw 128"; run "fusion_wgrad._SHORT_DW_TILE[0]=128"
echo "cut 6"; EXTRA="--fusion-cut 6" run
echo "base"; EXTRA="" run
echo "short dw 128"; run "fusion_wgrad._SHORT_DW_TILE[0]=128"
echo "cut 6"; EXTRA="--fusion-cut 6" run
