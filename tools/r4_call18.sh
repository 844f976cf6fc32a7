cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c18
timeout 1500 python -m pytest tests/test_attn_gpu.py tests/test_fusion_gpu.py tests/test_two_segment_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do
  (cd _r03 && BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>../gpurun_out/c18/t.err | cut -c62-105; echo "   [r03 code] $(grep 'GPU ms' ../gpurun_out/c18/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-200)")
  BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>gpurun_out/c18/t.err | cut -c62-105; echo "   [HEAD] $(grep 'GPU ms' gpurun_out/c18/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-200)"
done
