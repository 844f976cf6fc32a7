cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_graphed_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "plain_loop or module_api or fusion_backward_cut" 2>&1 | tail -30 > gpurun_out/c6/tests.log
cat gpurun_out/c6/tests.log | cut -c1-400
python bench.py --loop reference --steps 20 --warmup 5 2>gpurun_out/c6/ref.err | cut -c1-1200 > gpurun_out/c6/ref.json; cat gpurun_out/c6/ref.json; tail -3 gpurun_out/c6/ref.err
python bench.py --loop reference --graph off --steps 10 --warmup 3 2>gpurun_out/c6/refe.err | cut -c1-300 > gpurun_out/c6/refe.json; cat gpurun_out/c6/refe.json; tail -2 gpurun_out/c6/refe.err
