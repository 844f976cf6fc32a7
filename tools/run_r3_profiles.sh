# round-3 evidence in one gpurun call: step profile (kernel stats + one-step trace + phases), c3 / c5 / c2 bench lines,
# per-form GEMM PMC.  usage: bash tools/run_r3_profiles.sh <outdir-under-gpurun_out>
OUT=${1:-r3p}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
bash tools/run_gemm_pmc.sh $OUT/pmc > /dev/null 2>&1
cp gpurun_out/$OUT/pmc/gemm_pmc.jsonl gpurun_out/$OUT/gemm_pmc.jsonl; cp gpurun_out/$OUT/pmc/gemm_pmc_summary.txt gpurun_out/$OUT/gemm_pmc_summary.txt
cp gpurun_out/$OUT/gemm_pmc.jsonl profiles/r03_gemm_pmc.jsonl
bash tools/run_attn_pmc.sh $OUT/attn > /dev/null 2>&1
cp gpurun_out/$OUT/attn/attn_pmc_summary.txt profiles/r03_attn_pmc.txt
bash tools/run_step_profile.sh $OUT/step > gpurun_out/$OUT/step_profile.log 2>&1
cp gpurun_out/$OUT/step/kernel_stats.csv profiles/r03_c3_kernel_stats.csv
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$OUT/phases.err > /dev/null; grep -E "GPU ms|host ms" gpurun_out/$OUT/phases.err > gpurun_out/$OUT/c3_phases.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3.json 2> gpurun_out/$OUT/bench_c3.err; tail -c 400 gpurun_out/$OUT/bench_c3.json
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$OUT/bench_c2.json 2> gpurun_out/$OUT/bench_c2.err; head -c 300 gpurun_out/$OUT/bench_c2.json
python bench.py --workload c5 --steps 8 --warmup 2 > gpurun_out/$OUT/bench_c5.json 2> gpurun_out/$OUT/bench_c5.err; head -c 400 gpurun_out/$OUT/bench_c5.json; tail -3 gpurun_out/$OUT/bench_c5.err
export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT/c5prof -- python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/$OUT/c5prof.err; echo "c5 prof rc=$?"
cd $R
for f in $(find gpurun_out/$OUT/c5prof -name "*kernel_stats.csv"); do cp $f gpurun_out/$OUT/c5_kernel_stats.csv; done
rm -rf gpurun_out/$OUT/c5prof
head -12 gpurun_out/$OUT/c5_kernel_stats.csv | cut -c1-160
