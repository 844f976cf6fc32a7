# round-end verification: full GPU suite, smoke, default bench (with CPU baseline), c2 bench, phase timings, step profile
OUT=${1:-final}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 > gpurun_out/$OUT/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/$OUT/gpu_tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/$OUT/bench_c3.json 2> gpurun_out/$OUT/bench_c3.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/$OUT/bench_c3.json
timeout 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/$OUT/bench_c2.json 2> gpurun_out/$OUT/bench_c2.err; cut -c1-200 gpurun_out/$OUT/bench_c2.json
BQ_PIPE_TRACE=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep -E "GPU ms|host ms" | cut -c1-500 > gpurun_out/$OUT/phases.txt; cat gpurun_out/$OUT/phases.txt
timeout 300 python bench.py --no-cpu-baseline --graph off --steps 10 2>/dev/null | cut -c1-200 > gpurun_out/$OUT/bench_eager.json; cut -c1-200 gpurun_out/$OUT/bench_eager.json
bash tools/run_step_profile.sh $OUT/prof > /dev/null 2>&1; ls gpurun_out/$OUT/prof
