# the round-6 evidence that depends on the LAST code changes (gemm256's new K loop, the asm fragment reads, the grid build):
# per-form GEMM PMC, GEMM family vs hipBLASLt, ball-query PMC with every kernel, step profiles, phases, bench lines.  Attention /
# FPS / det-bwd PMC, the CPU baseline and the attention A/B files stay as tools/run_r6_profiles.sh left them (those kernels did
# not change).  usage: bash tools/run_r6_final.sh <outdir-under-gpurun_out>; then bash tools/collect_r6_final.sh <outdir> locally
OUT=${1:-r6f}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/$OUT
bash tools/run_gemm_pmc.sh $OUT/pmc > /dev/null 2>&1
cp gpurun_out/$OUT/pmc/gemm_pmc.jsonl gpurun_out/$OUT/gemm_pmc.jsonl; cp gpurun_out/$OUT/pmc/gemm_pmc_summary.txt gpurun_out/$OUT/gemm_pmc_summary.txt
cp gpurun_out/$OUT/gemm_pmc.jsonl profiles/r06_gemm_pmc.jsonl
python tools/bench_gemm2.py --json gpurun_out/$OUT/gemm_bench.json > gpurun_out/$OUT/gemm_bench.log 2>&1
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
