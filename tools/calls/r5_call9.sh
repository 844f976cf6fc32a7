cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
timeout 600 python tools/dbg_determinism.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | cut -c1-700
timeout 900 python -m pytest tests/test_attn_gpu.py -x -q --tb=short 2>&1 | tail -3
timeout 600 python tools/bench_attn.py 2>&1 | tail -6
BQ_ATTN_RAGGED_LAST=0 timeout 600 python tools/bench_attn.py 2>&1 | tail -6
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ragged last:', d['value'], d['ms_per_step'], d['roofline_attn']['fwd'], d['roofline_attn']['bwd'])"
BQ_ATTN_RAGGED_LAST=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2-D order:', d['value'], d['ms_per_step'], d['roofline_attn']['fwd'], d['roofline_attn']['bwd'])"
