run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>gpurun_out/e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; grep "host-side" gpurun_out/e.log; }
run BQ_DET_PRIORITY=0
run BQ_DET_PRIORITY=-1
run BQ_DET_PRIORITY=0 BQ_SCHEDULE=single
