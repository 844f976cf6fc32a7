cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c11
export TMPDIR=/tmp
timeout 1200 python -X faulthandler -m pytest tests/test_pipeline_gpu.py -x -v --tb=short > gpurun_out/c11/pipeline.log 2>&1; echo "pipeline rc=$?"
grep -v "Warning\|^  warn" gpurun_out/c11/pipeline.log | grep "PASSED\|FAILED\|ERROR\|Fatal\|File \"/root\|File \"/tmp/code" | tail -30 | cut -c1-220
timeout 1200 python -X faulthandler -m pytest tests/test_configs_gpu.py -x -v --tb=short > gpurun_out/c11/configs.log 2>&1; echo "configs rc=$?"
grep -v "Warning\|^  warn" gpurun_out/c11/configs.log | grep "PASSED\|FAILED\|ERROR\|Fatal\|File \"/root\|File \"/tmp/code" | tail -30 | cut -c1-220
BQ_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c11/bench_default.json 2> gpurun_out/c11/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c11/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
grep "GPU ms since\|reference loop" gpurun_out/c11/bench_default.err | cut -c1-1200
