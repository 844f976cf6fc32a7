cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "inverted or ball_query" --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -8
timeout 900 python -m pytest tests/test_modules_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -8
timeout 600 python tools/dbg_determinism.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | cut -c1-700
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deterministic scatter:', d['value'], d['ms_per_step'])"
python tools/ab_bench.py _ext.DETERMINISTIC_SCATTER[0]=False -- --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('atomic scatter:', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deterministic scatter:', d['value'], d['ms_per_step'])"
