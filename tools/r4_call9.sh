cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
timeout 900 python -m pytest tests/test_graphed_gpu.py -x -q -m gpu 2>&1 | tail -12 | cut -c1-600 > gpurun_out/c9/tests.log; cat gpurun_out/c9/tests.log
python bench.py --loop reference --steps 20 --warmup 5 2>gpurun_out/c9/ref.err | cut -c1-200; tail -1 gpurun_out/c9/ref.err
python bench.py --loop reference --no-wrap-loss --steps 20 --warmup 5 2>gpurun_out/c9/ref2.err | cut -c1-200; tail -1 gpurun_out/c9/ref2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c9/ph.err | cut -c1-200
BENCH_ARGS="--loop reference" bash tools/run_step_profile.sh c9/prof > gpurun_out/c9/prof.log 2>&1; tail -2 gpurun_out/c9/prof.log | cut -c1-200
