# kernel-trace + stats of a short bench run, one-step trace extracted (extra bench.py arguments: BENCH_ARGS="...").  usage: bash tools/run_step_profile.sh <outdir-under-gpurun_out> [env assignments...]
OUT=${1:-prof}; shift
export TMPDIR=/tmp "$@"
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT/prof -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/$OUT/prof_bench.json 2> $R/gpurun_out/$OUT/prof_bench.err; echo "prof rc=$?"
cd $R
for f in $(find gpurun_out/$OUT/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/$OUT/kernel_stats.csv; done
python tools/trace_tail.py "gpurun_out/$OUT/prof/*/*kernel_trace.csv" gpurun_out/$OUT/one_step_trace.csv 2
find gpurun_out/$OUT -name "*.csv" -size +30M -delete
find gpurun_out/$OUT -name "*.db" -delete
rm -rf gpurun_out/$OUT/prof
cut -c1-300 gpurun_out/$OUT/prof_bench.json
