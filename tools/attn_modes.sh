for m in "1 3" "3 2" "3 3" "3 4"; do set -- $m; echo "mode $1 minw $2"; BQ_ATTN_FWD_MODE=$1 BQ_ATTN_MINW=$2 python tools/bench_attn.py 2>&1 | grep "fwd" | grep -v torch | cut -c1-60; done
BQ_ATTN_FWD_MODE=3 python -m pytest tests/test_attn_gpu.py -x -q 2>&1 | tail -3
