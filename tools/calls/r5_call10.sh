cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "inverted" --tb=short 2>&1 | tail -3
timeout 600 python tools/dbg_determinism.py 2>&1 | grep -v "Warning\|warn\|amdgpu.ids\|Variable._exec" | cut -c1-900
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_modules_gpu.py -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -8
BQ_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c10/bench_default.json 2> gpurun_out/c10/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c10/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
grep "GPU ms since\|reference loop" gpurun_out/c10/bench_default.err | cut -c1-1200
