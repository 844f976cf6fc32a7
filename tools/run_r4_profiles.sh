# round-4 evidence in one gpurun call: per-form GEMM PMC, attention PMC, ball-query PMC (copied into profiles/ FIRST: bench.py
# reads them back), step profile (kernel stats + one-step trace + phases), c3 / c2 / c5 bench lines, the drop-in loop's lines.
# usage: bash tools/run_r4_profiles.sh <outdir-under-gpurun_out>
OUT=${1:-r4p}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/$OUT
bash tools/run_gemm_pmc.sh $OUT/pmc > /dev/null 2>&1
cp gpurun_out/$OUT/pmc/gemm_pmc.jsonl gpurun_out/$OUT/gemm_pmc.jsonl; cp gpurun_out/$OUT/pmc/gemm_pmc_summary.txt gpurun_out/$OUT/gemm_pmc_summary.txt
cp gpurun_out/$OUT/gemm_pmc.jsonl profiles/r04_gemm_pmc.jsonl
bash tools/run_attn_pmc.sh $OUT/attn > /dev/null 2>&1
cp gpurun_out/$OUT/attn/attn_pmc_summary.txt profiles/r04_attn_pmc.txt
bash tools/run_ballquery_pmc.sh $OUT/bq > gpurun_out/$OUT/bq.log 2>&1
cp gpurun_out/$OUT/bq/ballquery_pmc.json profiles/r04_ballquery_pmc.json
bash tools/run_step_profile.sh $OUT/step > gpurun_out/$OUT/step_profile.log 2>&1
cp gpurun_out/$OUT/step/kernel_stats.csv profiles/r04_c3_kernel_stats.csv
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$OUT/phases.err > /dev/null; grep -E "GPU ms|host ms" gpurun_out/$OUT/phases.err > gpurun_out/$OUT/c3_phases.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3.json 2> gpurun_out/$OUT/bench_c3.err; head -c 200 gpurun_out/$OUT/bench_c3.json; echo
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$OUT/bench_c2.json 2> gpurun_out/$OUT/bench_c2.err; head -c 200 gpurun_out/$OUT/bench_c2.json; echo
BQ_PIPE_TRACE=1 python bench.py --loop reference --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3_reference_loop.json 2> gpurun_out/$OUT/ref.err; head -c 200 gpurun_out/$OUT/bench_c3_reference_loop.json; echo; grep -E "GPU ms|host" gpurun_out/$OUT/ref.err > gpurun_out/$OUT/c3_reference_loop_phases.txt
python bench.py --loop reference --graph off --steps 8 --warmup 3 > gpurun_out/$OUT/bench_c3_reference_loop_eager.json 2> /dev/null; head -c 200 gpurun_out/$OUT/bench_c3_reference_loop_eager.json; echo
python bench.py --workload c5 --steps 8 --warmup 2 > gpurun_out/$OUT/bench_c5.json 2> gpurun_out/$OUT/bench_c5.err; head -c 200 gpurun_out/$OUT/bench_c5.json; echo; tail -2 gpurun_out/$OUT/bench_c5.err
