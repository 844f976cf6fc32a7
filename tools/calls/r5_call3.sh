cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_detloss_gpu.py tests/test_graphed_gpu.py "tests/test_modules_gpu.py" -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -40 > gpurun_out/c3/tests.log
tail -6 gpurun_out/c3/tests.log
BQ_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c3/bench_default.json 2> gpurun_out/c3/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c3/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
grep "GPU ms since\|reference loop" gpurun_out/c3/bench_default.err | cut -c1-1200
python tools/ab_bench.py loss_helper.FUSED_DET_LOSS[0]=False -- --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused loss:', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused loss:', d['value'], d['ms_per_step'])"
