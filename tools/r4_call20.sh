cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c20
run() { tag="$1"; shift; BQ_PIPE_TRACE=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>gpurun_out/c20/t.err | cut -c62-105; echo "   [$tag] $(grep 'GPU ms' gpurun_out/c20/t.err | sed 's/.*image_fwd/image_fwd/' | cut -c1-230)"; }
for i in 1 2; do
  run HEAD
  run cut6 --fusion-cut 6
  run cut9 --fusion-cut 9
  run noprologue --no-text-prologue
done
