cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c26
{
timeout 900 python -m pytest tests -x -q -m gpu -k "sharedmlp or point_major or wgrad or detector or c2 or c1 or pipeline or backbone" 2>&1 | tail -3
echo "== small gemm"; timeout 300 python tools/bench_small_gemm.py 2>&1 | grep "M="
run() { tag="$1"; dir="$2"; shift; shift; (cd $dir; BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>$GRAFT_REPO_ROOT/gpurun_out/c26/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' $GRAFT_REPO_ROOT/gpurun_out/c26/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-230)"); }
for i in 1 2; do
  run r03 _r03
  run HEAD .
done
} > gpurun_out/c26/log.txt 2>&1
cat gpurun_out/c26/log.txt
