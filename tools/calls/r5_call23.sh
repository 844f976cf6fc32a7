cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_modules_gpu.py -q -x 2>&1 | tail -3
for rep in 1 2; do
for v in "" pw1536 pw2048; do
  if [ -z "$v" ]; then unset BQHIP_LIB; else export BQHIP_LIB=$GRAFT_REPO_ROOT/bridgeqa_amd/lib/variants/libbqhip_$v.so; fi
  echo "== $v"; timeout 600 python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*"
done; done
unset BQHIP_LIB
BENCH_ARGS="--workload c2" bash tools/run_step_profile.sh r5c23/c2 > /dev/null 2>&1
