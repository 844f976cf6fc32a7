# round 4, GPU call 1: parity of the batched-row maps + the fusion goldens, then A/B of the concatenation-free K/V node
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
timeout 1500 python -m pytest tests/test_abi.py tests/test_gemm_gpu.py tests/test_fusion_gpu.py tests/test_two_segment_gpu.py tests/test_vit_cuts_cpu.py -x -q -m "gpu or not gpu" 2>&1 | tail -15 > gpurun_out/c1/tests1.log
cat gpurun_out/c1/tests1.log
for i in 1 2; do
  python tools/ab_bench.py "fusion_ops._TWIN_KV[0]=False" -- --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/c1/a$i.err | cut -c1-160 > gpurun_out/c1/a$i.json; cat gpurun_out/c1/a$i.json
  BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/c1/b$i.err | cut -c1-160 > gpurun_out/c1/b$i.json; cat gpurun_out/c1/b$i.json
  grep "GPU ms" gpurun_out/c1/b$i.err
done
timeout 1200 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/c1/tests2.log
cat gpurun_out/c1/tests2.log
