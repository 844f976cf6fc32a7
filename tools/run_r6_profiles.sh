# round-6 evidence in one gpurun call: per-form GEMM PMC, attention PMC, ball-query PMC (copied into profiles/ FIRST: bench.py
# reads them back), step profile (kernel stats + one-step trace + phases), c3 / c2 / c5 bench lines, the drop-in loop's lines.
# usage: bash tools/run_r6_profiles.sh <outdir-under-gpurun_out>
OUT=${1:-r6p}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/$OUT
bash tools/run_gemm_pmc.sh $OUT/pmc > /dev/null 2>&1
cp gpurun_out/$OUT/pmc/gemm_pmc.jsonl gpurun_out/$OUT/gemm_pmc.jsonl; cp gpurun_out/$OUT/pmc/gemm_pmc_summary.txt gpurun_out/$OUT/gemm_pmc_summary.txt
cp gpurun_out/$OUT/gemm_pmc.jsonl profiles/r06_gemm_pmc.jsonl
bash tools/run_attn_pmc.sh $OUT/attn > /dev/null 2>&1
cp gpurun_out/$OUT/attn/attn_pmc_summary.txt profiles/r06_attn_pmc.txt
bash tools/run_ballquery_pmc.sh $OUT/bq > gpurun_out/$OUT/bq.log 2>&1
cp gpurun_out/$OUT/bq/ballquery_pmc.json profiles/r06_ballquery_pmc.json
bash tools/run_det_bwd_pmc.sh $OUT/detbwd > gpurun_out/$OUT/detbwd.log 2>&1
cp gpurun_out/$OUT/detbwd/det_bwd_pmc_summary.txt profiles/r06_det_bwd_pmc.txt
BENCH_ARGS="--no-loop-reference" bash tools/run_step_profile.sh $OUT/step > gpurun_out/$OUT/step_profile.log 2>&1
cp gpurun_out/$OUT/step/kernel_stats.csv profiles/r06_c3_kernel_stats.csv
cp gpurun_out/$OUT/step/one_step_trace.csv profiles/r06_c3_one_step_trace.csv
BENCH_ARGS="--workload c2 --no-loop-reference" bash tools/run_step_profile.sh $OUT/c2step > gpurun_out/$OUT/c2step_profile.log 2>&1
cp gpurun_out/$OUT/c2step/kernel_stats.csv profiles/r06_c2_kernel_stats.csv
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/$OUT/phases.err > /dev/null; grep -E "GPU ms|host ms" gpurun_out/$OUT/phases.err > gpurun_out/$OUT/c3_phases.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3.json 2> gpurun_out/$OUT/bench_c3.err; head -c 200 gpurun_out/$OUT/bench_c3.json; echo
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$OUT/bench_c2.json 2> gpurun_out/$OUT/bench_c2.err; head -c 200 gpurun_out/$OUT/bench_c2.json; echo
BQ_PIPE_TRACE=1 python bench.py --loop reference --steps 20 --warmup 5 > gpurun_out/$OUT/bench_c3_reference_loop.json 2> gpurun_out/$OUT/ref.err; head -c 200 gpurun_out/$OUT/bench_c3_reference_loop.json; echo; grep -E "GPU ms|host" gpurun_out/$OUT/ref.err > gpurun_out/$OUT/c3_reference_loop_phases.txt
python bench.py --loop reference --graph off --steps 8 --warmup 3 > gpurun_out/$OUT/bench_c3_reference_loop_eager.json 2> /dev/null; head -c 200 gpurun_out/$OUT/bench_c3_reference_loop_eager.json; echo
python bench.py --workload c5 --steps 8 --warmup 2 > gpurun_out/$OUT/bench_c5.json 2> gpurun_out/$OUT/bench_c5.err; head -c 200 gpurun_out/$OUT/bench_c5.json; echo; tail -2 gpurun_out/$OUT/bench_c5.err
# FPS (SA1) counter passes: fresh traffic record for roofline_fps (replaces the round-1 citation)
export TMPDIR=/tmp
D=$R/gpurun_out/$OUT/fps
mkdir -p $D
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 $R/tools/fps_once.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/b -- python3 $R/tools/fps_once.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/c -- python3 $R/tools/fps_once.py > /dev/null 2>&1
cd $R
rm -f $D/fps.jsonl
python tools/pmc_summary.py $D/t $D/b $D/c --match fps_bucket --json $D/fps.jsonl --label fps_sa1 > $D/fps_pmc.txt 2>&1
python - <<PY
import json
rs = [json.loads(l) for l in open("$D/fps.jsonl")]
pick = lambda k: next((r[k] for r in rs if r.get(k) is not None), None)
rd, wr = pick("fetch_bytes"), pick("write_bytes")
out = {"B": 16, "N": 40000, "m": 2048, "avg_us": pick("avg_us"), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
       "bytes": (rd or 0) + (wr or 0), "file": "profiles/r06_fps_pmc.json",
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/fps_once.py; read = 2 x FETCH_SIZE x 1024 (gfx950), write = WRITE_SIZE x 1024; per launch"}
json.dump(out, open("$D/fps_pmc.json", "w")); print(out)
PY
cp $D/fps_pmc.json profiles/r06_fps_pmc.json; cp $D/fps_pmc.txt gpurun_out/$OUT/fps_pmc.txt
rm -rf $D/t $D/b $D/c

# round 6 extras: launch floor, attention A/B (alone and in the step), GEMM family vs hipBLASLt, the planned CPU baseline
cd $R
timeout 120 tools/probes/launch_floor > gpurun_out/$OUT/launch_floor.txt 2>&1
python tools/ab_attn.py 0 1 7 2>&1 | grep -v amdgpu.ids > gpurun_out/$OUT/ab_attn.txt
bash tools/ab_step_attn.sh > /dev/null 2>&1; cp gpurun_out/r6d/ab_step_attn.txt gpurun_out/$OUT/ab_step_attn.txt
python tools/bench_gemm2.py --json gpurun_out/$OUT/gemm_bench.json > gpurun_out/$OUT/gemm_bench.log 2>&1
for cfg in "0 16" "32 16" "32 2"; do timeout 1500 python tools/cpu_baseline_full.py $cfg 2>/dev/null | tail -1; done > gpurun_out/$OUT/cpu_baseline_full.jsonl
