#!/bin/bash
# A/B of the compile-time PLAIN attention paths (tools/bench_attn.py; ViT shape of config c3 and the c5 shape)
for mode in 0 1 2; do
  for w in 2 3; do
    echo "fwd mode=$mode minw=$w"
    BQ_ATTN_FWD_MODE=$mode BQ_ATTN_MINW=$w python tools/bench_attn.py 2>&1 | grep "^B=" | cut -c1-60
  done
done
echo "bwd generic"; BQ_ATTN_NO_PLAIN=1 python tools/bench_attn.py 2>&1 | grep "^B=" | cut -c1-140
for dq in 2 3; do for dkv in 1 2; do
  echo "bwd plain dq=$dq dkv=$dkv"; BQ_ATTN_DQ_MINW=$dq BQ_ATTN_DKV_MINW=$dkv python tools/bench_attn.py 2>&1 | grep "^B=" | cut -c1-140
done; done
