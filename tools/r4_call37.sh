cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c37
{
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python "$@" 2>gpurun_out/c37/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c37/t.err | sed 's/.*geometry/geometry/' | cut -c1-260)"; }
for i in 1 2; do
  run HEAD bench.py --steps 30 --warmup 5 --no-cpu-baseline
  run detprio-1 tools/ab_bench.py pipeline._DET_PRIORITY[0]=-1 -- --steps 30 --warmup 5 --no-cpu-baseline
done
} > gpurun_out/c37/log.txt 2>&1
cat gpurun_out/c37/log.txt
