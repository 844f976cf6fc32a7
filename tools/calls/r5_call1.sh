cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_graphed_gpu.py "tests/test_fusion_gpu.py::test_grad_sink_fails_loudly_when_a_level_never_runs_its_backward" tests/test_fusion_gpu.py -k "graphed or grad_sink or twin_kv or transposed" -x -q --tb=short 2>&1 | grep -v "Warning\|^  warn" | tail -40 > gpurun_out/c1/tests.log
tail -5 gpurun_out/c1/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c1/bench_default.json 2> gpurun_out/c1/bench_default.err
tail -c 600 gpurun_out/c1/bench_default.json; tail -5 gpurun_out/c1/bench_default.err
BQ_PIPE_TRACE=1 timeout 600 python bench.py --loop reference --steps 20 --warmup 5 > gpurun_out/c1/loop_ref.json 2> gpurun_out/c1/loop_ref.err
cat gpurun_out/c1/loop_ref.json | cut -c1-400; grep "reference loop" gpurun_out/c1/loop_ref.err
BQ_PIPE_TRACE=1 timeout 600 python bench.py --loop reference --no-prefetch --eager-optimizer --steps 20 --warmup 5 > gpurun_out/c1/loop_ref_r4style.json 2> gpurun_out/c1/loop_ref_r4style.err
cut -c1-200 gpurun_out/c1/loop_ref_r4style.json; grep "reference loop" gpurun_out/c1/loop_ref_r4style.err
timeout 900 python tools/microbatch_probe.py > gpurun_out/c1/microbatch.json 2> gpurun_out/c1/microbatch.err
cat gpurun_out/c1/microbatch.json; tail -3 gpurun_out/c1/microbatch.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c1/trace_ref -- python3 $GRAFT_REPO_ROOT/bench.py --loop reference --steps 3 --warmup 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/c1/trace_ref/*/ | head
python tools/trace_tail.py --help 2>&1 | head -5
python tools/trace_tail.py 'gpurun_out/c1/trace_ref/*/*_kernel_trace.csv' gpurun_out/c1/ref_loop_one_step_trace.csv 1
rm -rf gpurun_out/c1/trace_ref
