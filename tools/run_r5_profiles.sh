# round-5 evidence in one gpurun call: per-form GEMM PMC, attention PMC, ball-query PMC (copied into profiles/ FIRST: bench.py
# reads them back), step profile (kernel stats + one-step trace + phases), c3 / c2 / c5 bench lines, the drop-in loop's lines.
# usage: bash tools/run_r5_profiles.sh <outdir-under-gpurun_out>
OUT=${1:-r4p}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p $R/gpurun_out/$OUT
bash tools/run_gemm_pmc.sh $OUT/pmc > /dev/null 2>&1
cp gpurun_out/$OUT/pmc/gemm_pmc.jsonl gpurun_out/$OUT/gemm_pmc.jsonl; cp gpurun_out/$OUT/pmc/gemm_pmc_summary.txt gpurun_out/$OUT/gemm_pmc_summary.txt
cp gpurun_out/$OUT/gemm_pmc.jsonl profiles/r05_gemm_pmc.jsonl
bash tools/run_attn_pmc.sh $OUT/attn > /dev/null 2>&1
cp gpurun_out/$OUT/attn/attn_pmc_summary.txt profiles/r05_attn_pmc.txt
bash tools/run_ballquery_pmc.sh $OUT/bq > gpurun_out/$OUT/bq.log 2>&1
cp gpurun_out/$OUT/bq/ballquery_pmc.json profiles/r05_ballquery_pmc.json
bash tools/run_det_bwd_pmc.sh $OUT/detbwd > gpurun_out/$OUT/detbwd.log 2>&1
cp gpurun_out/$OUT/detbwd/det_bwd_pmc_summary.txt profiles/r05_det_bwd_pmc.txt
bash tools/run_step_profile.sh $OUT/step > gpurun_out/$OUT/step_profile.log 2>&1
cp gpurun_out/$OUT/step/kernel_stats.csv profiles/r05_c3_kernel_stats.csv
cp gpurun_out/$OUT/step/one_step_trace.csv profiles/r05_c3_one_step_trace.csv
BENCH_ARGS="--workload c2" bash tools/run_step_profile.sh $OUT/c2step > gpurun_out/$OUT/c2step_profile.log 2>&1
cp gpurun_out/$OUT/c2step/kernel_stats.csv profiles/r05_c2_kernel_stats.csv
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
       "bytes": (rd or 0) + (wr or 0), "file": "profiles/r05_fps_pmc.json",
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/fps_once.py; read = 2 x FETCH_SIZE x 1024 (gfx950), write = WRITE_SIZE x 1024; per launch"}
json.dump(out, open("$D/fps_pmc.json", "w")); print(out)
PY
cp $D/fps_pmc.json profiles/r05_fps_pmc.json; cp $D/fps_pmc.txt gpurun_out/$OUT/fps_pmc.txt
rm -rf $D/t $D/b $D/c
