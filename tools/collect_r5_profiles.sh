# copies what tools/run_r5_profiles.sh left under gpurun_out/<OUT> into profiles/r05_* (run locally after the gpurun call)
OUT=${1:-r5p}
G=gpurun_out/$OUT
cp $G/gemm_pmc.jsonl profiles/r05_gemm_pmc.jsonl
cp $G/gemm_pmc_summary.txt profiles/r05_gemm_pmc.txt
cp $G/attn/attn_pmc_summary.txt profiles/r05_attn_pmc.txt
cp $G/bq/ballquery_pmc.json profiles/r05_ballquery_pmc.json
cp $G/detbwd/det_bwd_pmc_summary.txt profiles/r05_det_bwd_pmc.txt
cp $G/step/kernel_stats.csv profiles/r05_c3_kernel_stats.csv
cp $G/step/one_step_trace.csv profiles/r05_c3_one_step_trace.csv
cp $G/c2step/kernel_stats.csv profiles/r05_c2_kernel_stats.csv
cp $G/c3_phases.txt profiles/r05_c3_phases.txt
cp $G/bench_c3.json profiles/r05_bench_c3.json
cp $G/bench_c2.json profiles/r05_bench_c2.json
cp $G/bench_c5.json profiles/r05_bench_c5.json
cp $G/bench_c3_reference_loop.json profiles/r05_bench_c3_reference_loop.json
cp $G/bench_c3_reference_loop_eager.json profiles/r05_bench_c3_reference_loop_eager.json
cp $G/c3_reference_loop_phases.txt profiles/r05_c3_reference_loop_phases.txt
cp $G/fps/fps_pmc.json profiles/r05_fps_pmc.json
cp $G/fps_pmc.txt profiles/r05_fps_pmc.txt
ls -la profiles/r05_*
