for dq in 2 3; do for dkv in 1 2; do echo "dq=$dq dkv=$dkv"; BQ_ATTN_DQ_MINW=$dq BQ_ATTN_DKV_MINW=$dkv timeout 120 python tools/bench_attn.py 2>&1 | grep "L=1025" | cut -c1-120; done; done
for w in 2 3; do echo "fwd minw=$w"; BQ_ATTN_MINW=$w timeout 120 python tools/bench_attn.py 2>&1 | grep "L=1025" | cut -c1-60; done
