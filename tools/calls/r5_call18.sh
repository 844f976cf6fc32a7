cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5c18; mkdir -p $O
timeout 900 python -m pytest tests/test_modules_gpu.py -q -x 2>&1 | tail -5
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline 2>$O/c2.err | cut -c80-180
timeout 900 python bench.py --steps 30 --warmup 5 --no-loop-reference --no-cpu-baseline 2>$O/c3.err | cut -c80-180
BENCH_ARGS="--workload c2" bash tools/run_step_profile.sh r5c18/c2 > $O/c2_profile.log 2>&1
