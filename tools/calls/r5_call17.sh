cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=r5c17; mkdir -p gpurun_out/$O
BENCH_ARGS="--workload c2" bash tools/run_step_profile.sh $O/c2 > gpurun_out/$O/c2_profile.log 2>&1
for i in 1 2; do
BQ_FUSED_SA_BWD=0 timeout 900 python bench.py --steps 30 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | cut -c80-180
BQ_FUSED_SA_BWD=1 timeout 900 python bench.py --steps 30 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | cut -c80-180
done
