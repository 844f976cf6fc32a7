# copies what tools/run_r6_final.sh left under gpurun_out/<OUT> into profiles/r06_* (run locally after the gpurun call)
OUT=${1:-r6f}
G=gpurun_out/$OUT
cp $G/gemm_pmc.jsonl profiles/r06_gemm_pmc.jsonl
cp $G/gemm_pmc_summary.txt profiles/r06_gemm_pmc.txt
cp $G/gemm_bench.json profiles/r06_gemm_bench.json
cp $G/bq/ballquery_pmc.json profiles/r06_ballquery_pmc.json
cp $G/step/kernel_stats.csv profiles/r06_c3_kernel_stats.csv
cp $G/step/one_step_trace.csv profiles/r06_c3_one_step_trace.csv
cp $G/c2step/kernel_stats.csv profiles/r06_c2_kernel_stats.csv
cp $G/c3_phases.txt profiles/r06_c3_phases.txt
cp $G/bench_c3.json profiles/r06_bench_c3.json
cp $G/bench_c2.json profiles/r06_bench_c2.json
cp $G/bench_c5.json profiles/r06_bench_c5.json
cp $G/bench_c3_reference_loop.json profiles/r06_bench_c3_reference_loop.json
cp $G/bench_c3_reference_loop_eager.json profiles/r06_bench_c3_reference_loop_eager.json
cp $G/c3_reference_loop_phases.txt profiles/r06_c3_reference_loop_phases.txt
ls -la profiles/r06_* | head -40
