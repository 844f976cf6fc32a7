# the c3 step under every bq_attn_set_persistent mask (0: one workgroup per block; 1: resident forward; 5: + dK/dV; 7: all three),
# one process per run, A/B/../A order inside ONE gpurun call -> gpurun_out/r6d/ab_step_attn.txt
mkdir -p gpurun_out/r6d
for m in 0 1 5 7 0 7; do
  python tools/ab_bench.py "call:_ext.attn_set_persistent($m)" -- --steps 30 --warmup 8 --no-cpu-baseline --no-loop-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask $m', d['ms_per_step'])"
done | tee gpurun_out/r6d/ab_step_attn.txt
