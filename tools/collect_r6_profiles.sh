# copies what tools/run_r6_profiles.sh left under gpurun_out/<OUT> into profiles/r06_* (run locally after the gpurun call)
OUT=${1:-r6p}
G=gpurun_out/$OUT
cp $G/gemm_pmc.jsonl profiles/r06_gemm_pmc.jsonl
cp $G/gemm_pmc_summary.txt profiles/r06_gemm_pmc.txt
cp $G/attn/attn_pmc_summary.txt profiles/r06_attn_pmc.txt
cp $G/bq/ballquery_pmc.json profiles/r06_ballquery_pmc.json
cp $G/detbwd/det_bwd_pmc_summary.txt profiles/r06_det_bwd_pmc.txt
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
cp $G/fps/fps_pmc.json profiles/r06_fps_pmc.json
cp $G/fps_pmc.txt profiles/r06_fps_pmc.txt
ls -la profiles/r06_*
cp $G/launch_floor.txt profiles/r06_launch_floor.txt
cp $G/ab_attn.txt profiles/r06_attn_ab.txt
cp $G/ab_step_attn.txt profiles/r06_attn_ab_in_step.txt
cp $G/gemm_bench.json profiles/r06_gemm_bench.json 2>/dev/null
python - <<PY
import json
rows=[json.loads(l) for l in open("$G/cpu_baseline_full.jsonl") if l.strip().startswith("{")]
json.dump({"what": "BASELINE.md section 2's planned CPU baseline (c3 hot path over the CPU oracle, fwd+bwd+clip+AdamW): all cores x 16 scenes, 32 threads x 16 scenes, 32 threads x 2 scenes (the default bench line's bounded sample); tools/cpu_baseline_full.py", "runs": rows}, open("profiles/r06_cpu_baseline_full.json","w"), indent=1)
print(rows)
PY
ls -la profiles/r06_*
