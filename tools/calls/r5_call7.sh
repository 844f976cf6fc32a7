cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_graphed_gpu.py tests/test_detloss_gpu.py tests/test_configs_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -15
BQ_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bench_default.json 2> gpurun_out/c7/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c7/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
grep "GPU ms since\|reference loop" gpurun_out/c7/bench_default.err | cut -c1-1200
python tools/ab_bench.py _ext.FP32_PREACT[0]=False -- --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stored pre-activation apply:', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32-accumulator apply:', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2:', d['value'], d['ms_per_step'])"
timeout 1500 python tools/loss_gap_r5.py --steps 200 --reps 3 --out gpurun_out/c7/loss_gap.json 2>&1 | grep -v "Warning\|warn" | tail -12
