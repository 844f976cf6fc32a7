cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" satwo pw768 pw1024; do
  if [ -z "$v" ]; then unset BQHIP_LIB; else export BQHIP_LIB=$GRAFT_REPO_ROOT/bridgeqa_amd/lib/variants/libbqhip_$v.so; fi
  echo "== $v"; timeout 600 python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*"
done; done
