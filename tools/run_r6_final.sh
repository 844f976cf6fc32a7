# the part of tools/run_r6_profiles.sh that depends on the LAST code changes of round 6 (LayerNorm backward from the stored
# sum, grid build): step profiles, phases, bench lines, ball-query PMC with BOTH kernels.  GEMM / attention / FPS / det-bwd PMC,
# the CPU baseline and the A/B files stay as run_r6_profiles.sh left them (those kernels did not change).
OUT=${1:-r6f}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/$OUT
bash tools/run_ballquery_pmc.sh $OUT/bq > gpurun_out/$OUT/bq.log 2>&1
cp gpurun_out/$OUT/bq/ballquery_pmc.json profiles/r06_ballquery_pmc.json
BENCH_ARGS="--no-loop-reference" bash tools/run_step_profile.sh $OUT/step > gpurun_out/$OUT/step_profile.log 2>&1
cp gpurun_out/$OUT/step/kernel_stats.csv profiles/r06_c3_kernel_stats.csv
BENCH_ARGS="--workload c2 --no-loop-reference" bash tools/run_step_profile.sh $OUT/c2step > gpurun_out/$OUT/c2step_profile.log 2>&1
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$OUT/phases.err > /dev/null; grep -E "GPU ms|host ms" gpurun_out/$OUT/phases.err > gpurun_out/$OUT/c3_phases.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3.json 2> gpurun_out/$OUT/bench_c3.err; head -c 200 gpurun_out/$OUT/bench_c3.json; echo
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$OUT/bench_c2.json 2> gpurun_out/$OUT/bench_c2.err; head -c 200 gpurun_out/$OUT/bench_c2.json; echo
BQ_PIPE_TRACE=1 python bench.py --loop reference --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3_reference_loop.json 2> gpurun_out/$OUT/ref.err; head -c 200 gpurun_out/$OUT/bench_c3_reference_loop.json; echo; grep -E "GPU ms|host" gpurun_out/$OUT/ref.err > gpurun_out/$OUT/c3_reference_loop_phases.txt
python bench.py --loop reference --graph off --steps 8 --warmup 3 > gpurun_out/$OUT/bench_c3_reference_loop_eager.json 2> /dev/null
python bench.py --workload c5 --steps 8 --warmup 2 > gpurun_out/$OUT/bench_c5.json 2> gpurun_out/$OUT/bench_c5.err; head -c 200 gpurun_out/$OUT/bench_c5.json; echo
