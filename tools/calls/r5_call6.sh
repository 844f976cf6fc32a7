cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
timeout 600 python tools/dbg_graphed_ddp.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | cut -c1-700
timeout 600 python tools/dbg_detloss.py 2>&1 | grep "fused_ok\|loss "
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now:', d['value'], d['ms_per_step'])"
python tools/ab_bench.py loss_helper.FUSED_DET_LOSS[0]=False -- --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused loss:', d['value'], d['ms_per_step'])"
