cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
timeout 600 python tools/dbg_graphed_ddp.py > gpurun_out/c5/dbg_ddp.log 2>&1; grep "second enable\|forced exchange\|Error\|error" gpurun_out/c5/dbg_ddp.log | cut -c1-900
timeout 600 python tools/dbg_detloss.py 2>&1 | grep -v "Warning\|^  warn\|amdgpu.ids" | tail -40
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "ball_query" --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -12
timeout 600 python tools/time_ball_query.py 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "stream_k" --tb=short 2>&1 | tail -3
timeout 900 python tools/bench_gemm2.py --big 2>&1 | grep "fc2+bias\|dx qkv\|dx fc1 "
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamK on + grid bq:', d['value'], d['ms_per_step'], d['roofline_ballquery'])"
BQ_GEMM_STREAMK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamK off + grid bq:', d['value'], d['ms_per_step'])"
BQ_GEMM_STREAMK=0 python tools/ab_bench.py _ext.BALL_QUERY_GRID_MIN_N[0]=1073741824 -- --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamK off, scan bq:', d['value'], d['ms_per_step'])"
