run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>gpurun_out/e.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; tail -1 gpurun_out/e.log | cut -c1-150; }
run BQ_RESERVE_CUS=0
run BQ_RESERVE_CUS=16
run BQ_RESERVE_CUS=32
run BQ_RESERVE_CUS=48
