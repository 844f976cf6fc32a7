run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep "host-side" gpurun_out/e.log; }
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=5
run GPU_MAX_HW_QUEUES=6
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
