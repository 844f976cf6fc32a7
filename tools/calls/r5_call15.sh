cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=r5c15; mkdir -p gpurun_out/$O
BENCH_ARGS="--workload c2" bash tools/run_step_profile.sh $O/c2 > gpurun_out/$O/c2_profile.log 2>&1
tail -2 gpurun_out/$O/c2_profile.log | cut -c1-300
