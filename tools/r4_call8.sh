cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
python - <<'PY' > gpurun_out/c8/control.log 2>&1
import sys, torch
sys.path.insert(0, "tests")
import test_graphed_gpu as t
from bridgeqa_amd import fusion_ops as ops
ops.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda:0")
a, la = t._grads_once(dev, "eager")
b, lb = t._grads_once(dev, "eager")
c, lc = t._grads_once(dev, "graphed")
d, ld = t._grads_once(dev, "wrapped")
def worst(x, y):
    def err(n):
        ref = y[n.replace(".key.bias", ".value.bias")] if n.endswith(".key.bias") else y[n]
        return ((x[n] - y[n]).norm() / (ref.norm() + 1e-12)).item()
    return sorted(((err(n), n) for n in y), reverse=True)[:5]
print("losses", la, lb, lc, ld)
print("eager vs eager  ", worst(b, a))
print("graphed vs eager", worst(c, a))
print("wrapped vs eager", worst(d, a))
print("wrapped vs graphed", worst(d, c))
PY
cat gpurun_out/c8/control.log | tail -8
python bench.py --loop reference --steps 20 --warmup 5 2>gpurun_out/c8/ref.err | cut -c1-200; tail -1 gpurun_out/c8/ref.err
python bench.py --loop reference --no-wrap-loss --steps 20 --warmup 5 2>gpurun_out/c8/ref2.err | cut -c1-200; tail -1 gpurun_out/c8/ref2.err
BENCH_ARGS="--loop reference" bash tools/run_step_profile.sh c8/prof > gpurun_out/c8/prof.log 2>&1; tail -2 gpurun_out/c8/prof.log | cut -c1-300
