cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
BQ_PIPE_TRACE=1 python bench.py --loop reference --steps 20 --warmup 5 2>gpurun_out/c10/ref.err | cut -c1-200; tail -3 gpurun_out/c10/ref.err
BQ_PIPE_TRACE=1 python tools/ab_bench.py "pipeline._SINGLE_STREAM_GRAPHS[0]=True" -- --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c10/ph1.err | cut -c1-200; grep "host ms\|GPU ms" gpurun_out/c10/ph1.err
BQ_PIPE_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c10/ph0.err | cut -c1-200; grep "host ms\|GPU ms" gpurun_out/c10/ph0.err
BQ_PIPE_TRACE=1 python tools/ab_bench.py "pipeline._SINGLE_STREAM_GRAPHS[0]=True" -- --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c10/ph1.err | cut -c1-200; grep "host ms\|GPU ms" gpurun_out/c10/ph1.err
