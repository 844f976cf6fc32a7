cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=r5c14; mkdir -p gpurun_out/$O
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop-reference 2> gpurun_out/$O/phases.err | cut -c1-200
grep -E "GPU ms|host ms" gpurun_out/$O/phases.err | tail -30
bash tools/run_step_profile.sh $O/step > gpurun_out/$O/step_profile.log 2>&1
ls gpurun_out/$O/step
