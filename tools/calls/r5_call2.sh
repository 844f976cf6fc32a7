cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_graphed_gpu.py tests/test_fusion_gpu.py -k "graphed or grad_sink or twin_kv or transposed or prefetch or wrapped or enable" -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -40 > gpurun_out/c2/tests.log
tail -5 gpurun_out/c2/tests.log
BQ_PIPE_TRACE=1 timeout 600 python bench.py --loop reference --steps 20 --warmup 5 > gpurun_out/c2/loop_ref.json 2> gpurun_out/c2/loop_ref.err
cut -c1-300 gpurun_out/c2/loop_ref.json; grep "reference loop" gpurun_out/c2/loop_ref.err
BQ_PIPE_TRACE=1 timeout 600 python bench.py --loop reference --no-prefetch --eager-optimizer --steps 20 --warmup 5 > gpurun_out/c2/loop_ref_r4style.json 2> gpurun_out/c2/loop_ref_r4style.err
cut -c1-200 gpurun_out/c2/loop_ref_r4style.json; grep "reference loop" gpurun_out/c2/loop_ref_r4style.err
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c2/bench_default.json 2> gpurun_out/c2/bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/c2/bench_default.json')); print(d['value'], d['ms_per_step'], d.get('loop_reference'))"
