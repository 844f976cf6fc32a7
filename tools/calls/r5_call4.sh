cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
timeout 600 python tools/dbg_graphed_ddp.py 2>&1 | grep -v "Warning\|^  warn\|amdgpu.ids" | tail -5
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "stream_k or transposed or 256x128 or forward_bias" --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -15
timeout 900 python -m pytest tests/test_detloss_gpu.py -x -q --tb=short 2>&1 | tail -3
timeout 900 python tools/bench_gemm2.py --big --json gpurun_out/c4/gemm_bench.json 2>&1 | tail -16
BQ_PIPE_TRACE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c4/bench_default.json 2> gpurun_out/c4/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c4/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
grep "GPU ms since\|reference loop" gpurun_out/c4/bench_default.err | cut -c1-1200
BQ_GEMM_STREAMK=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamK off:', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-loop-reference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streamK on:', d['value'], d['ms_per_step'])"
