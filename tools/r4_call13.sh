cd $GRAFT_REPO_ROOT
for w in twin decoder; do echo "== $w"; timeout 300 python tools/bench_short_dw.py $w 2>&1 | tail -4; done
