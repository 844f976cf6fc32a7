cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c19; mkdir -p $O
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop-reference 2> $O/phases.err | cut -c80-180
grep -E "GPU ms|host ms" $O/phases.err | tail -3
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/c2.err | cut -c80-180
